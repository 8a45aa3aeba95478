// Input generation only (bench.py / tests): the `std::mt19937{seed}` byte stream of the reference
// benchmarks -- benchmark/multi_commitment/benchmark.m.cc:141-156 draws every scalar byte from
// std::uniform_int_distribution<uint8_t> on one std::mt19937, which in libstdc++ keeps the top 8 bits
// of each 32-bit output (Lemire's multiply-shift with a power-of-two range never rejects) -- produced
// on many host threads at once.  BASELINE configs 4 and 5 need 2^33 and 3.3e9 draws of that ONE
// serial stream (215 s / 83 s through numpy, ~25 s in a tight loop); here thread j starts from the
// generator state after j * L draws, obtained by polynomial jump-ahead:
//   * the transition F of the 19937-bit state is linear over GF(2); its characteristic polynomial phi
//     (degree 19937) is recovered with Berlekamp-Massey from one output bit of 2 x 19937 steps;
//   * F^J s = g(F) s with g = x^J mod phi, evaluated by Horner's rule in 19937 state steps
//     (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random
//     number generators", 2008 -- the published algorithm, written from its description).
// Nothing here is part of the product.
//
//   g++ -O2 -std=c++17 -fPIC -shared -pthread tools/mt19937/mtstream.cc -o tools/mt19937/_build/libmtstream.so
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {
constexpr int N = 624, M = 397, MEXP = 19937;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

// the sequence x_0, x_1, ... as a sliding window of 624 words: w[(head + i) % N] = x_{t + i}
struct window {
  uint32_t w[N];
  int head = 0;
  void seed(uint32_t s) { // init_genrand, == std::mt19937{s}
    w[0] = s;
    for (int i = 1; i < N; ++i) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + static_cast<uint32_t>(i);
    head = 0;
  }
  // x_{t + 624} from x_t, x_{t + 1}, x_{t + 397}; the window moves on by one word
  uint32_t step() {
    uint32_t& x0 = w[head];
    const uint32_t x1 = w[head + 1 == N ? 0 : head + 1];
    const uint32_t xm = w[head + M >= N ? head + M - N : head + M];
    const uint32_t y = (x0 & UPPER) | (x1 & LOWER);
    x0 = xm ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    const uint32_t fresh = x0;
    head = head + 1 == N ? 0 : head + 1;
    return fresh;
  }
  void add(const window& o) { // aligned XOR
    for (int i = 0; i < N; ++i) w[(head + i) % N] ^= o.w[(o.head + i) % N];
  }
};

inline uint32_t temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// polynomials over GF(2), bit i = coefficient of x^i
using poly = std::vector<uint64_t>;
constexpr int PW = (2 * MEXP + 63) / 64 + 1;
inline bool bit(const poly& p, int i) { return (p[i >> 6] >> (i & 63)) & 1u; }
inline void flip(poly& p, int i) { p[i >> 6] ^= uint64_t{1} << (i & 63); }
// p ^= q << s  (q has `qwords` significant words)
void xor_shifted(poly& p, const poly& q, int qwords, int s) {
  const int ws = s >> 6, bs = s & 63, size = static_cast<int>(p.size());
  for (int i = 0; i < qwords && i + ws < size; ++i) {
    if (q[i] == 0) continue;
    p[i + ws] ^= q[i] << bs;
    if (bs != 0 && i + ws + 1 < size) p[i + ws + 1] ^= q[i] >> (64 - bs);
  }
}

// characteristic polynomial of the transition: Berlekamp-Massey on the low bit of x_t
const poly& characteristic() {
  static poly phi;
  static std::once_flag once;
  std::call_once(once, [] {
    const int T = 2 * MEXP + 64;
    std::vector<uint8_t> s(T);
    window g;
    g.seed(5489u);
    for (int i = 0; i < T; ++i) s[i] = g.step() & 1u;
    // connection polynomials C, B (bit i = c_i), current length L; standard GF(2) BM with the
    // discrepancy as the parity of (C AND the reversed sequence window)
    const int W = (T + 63) / 64 + 2;
    poly C(W, 0), B(W, 0), tmp;
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    // rev[k] holds s[n - k] for the current n, kept as a bitset that is shifted up by one per step
    poly rev(W, 0);
    for (int n = 0; n < T; ++n) {
      // rev <<= 1; rev bit 0 = s[n]
      uint64_t carry = s[n];
      for (int i = 0; i < W; ++i) {
        const uint64_t next = rev[i] >> 63;
        rev[i] = (rev[i] << 1) | carry;
        carry = next;
      }
      uint64_t acc = 0;
      const int words = (L >> 6) + 1;
      for (int i = 0; i < words; ++i) acc ^= C[i] & rev[i];
      const bool d = __builtin_parityll(acc);
      if (!d) {
        ++m;
      } else if (2 * L <= n) {
        tmp = C;
        xor_shifted(C, B, W, m);
        L = n + 1 - L;
        B = tmp;
        m = 1;
      } else {
        xor_shifted(C, B, W, m);
        ++m;
      }
    }
    // phi(x) = x^L C(1 / x): coefficient of x^(L - i) = c_i
    phi.assign(PW, 0);
    for (int i = 0; i <= L; ++i) {
      if (bit(C, i)) flip(phi, L - i);
    }
    if (L != MEXP) phi.clear(); // cannot happen for MT19937; callers fall back to the serial path
  });
  return phi;
}

// r = r mod phi for deg r < 2 MEXP
void reduce(poly& r, const poly& phi) {
  const int pw = (MEXP >> 6) + 1;
  for (int i = 2 * MEXP - 1; i >= MEXP; --i) {
    if (bit(r, i)) xor_shifted(r, phi, pw, i - MEXP);
  }
}

// x^J mod phi
poly power_of_x(uint64_t J, const poly& phi) {
  poly r(PW, 0), sq(PW, 0);
  r[0] = 1;
  for (int b = 63; b >= 0; --b) {
    // square: spread the bits
    std::fill(sq.begin(), sq.end(), 0);
    for (int i = 0; i < MEXP; ++i) {
      if (bit(r, i)) flip(sq, 2 * i);
    }
    reduce(sq, phi);
    r.swap(sq);
    if ((J >> b) & 1u) {
      // times x
      uint64_t carry = 0;
      for (int i = 0; i < PW; ++i) {
        const uint64_t next = r[i] >> 63;
        r[i] = (r[i] << 1) | carry;
        carry = next;
      }
      if (bit(r, MEXP)) xor_shifted(r, phi, (MEXP >> 6) + 1, 0);
    }
  }
  return r;
}

// s <- g(F) s  (Horner: the top coefficient first)
void jump(window& s, const poly& g) {
  int i = MEXP - 1;
  while (i >= 0 && !bit(g, i)) --i;
  if (i < 0) {
    std::memset(s.w, 0, sizeof(s.w));
    return;
  }
  window t = s;
  for (--i; i >= 0; --i) {
    t.step();
    if (bit(g, i)) t.add(s);
  }
  s = t;
}

void generate(uint8_t* out, uint64_t count, window g, unsigned shift) {
  for (uint64_t i = 0; i < count; ++i) out[i] = static_cast<uint8_t>(temper(g.step()) >> shift);
}
} // namespace

extern "C" {
// out[i] = draw i of uniform_int_distribution<uint8_t>{0, 255} (shift 24; {0, 1}: shift 31) on
// std::mt19937{seed}, for skip <= i < skip + count; `threads` host threads (0: hardware
// concurrency), each on a contiguous piece of at least 2^20 draws.  Returns the threads used.
int mt19937_fill(uint8_t* out, uint64_t count, uint32_t seed, uint64_t skip, unsigned threads,
                 unsigned shift) {
  if (count == 0) return 0;
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  const uint64_t least = uint64_t{1} << 20;
  uint64_t pieces = std::min<uint64_t>(threads, (count + least - 1) / least);
  const poly* phi = nullptr;
  if (pieces > 1 || skip != 0) {
    phi = &characteristic();
    if (phi->empty()) { // (defensive) no jump available: one serial pass
      pieces = 1;
      phi = nullptr;
    }
  }
  window first;
  first.seed(seed);
  if (skip != 0) {
    if (phi != nullptr) {
      jump(first, power_of_x(skip, *phi));
    } else {
      for (uint64_t i = 0; i < skip; ++i) first.step();
    }
  }
  const uint64_t piece = (count + pieces - 1) / pieces;
  std::vector<window> starts(pieces);
  starts[0] = first;
  if (pieces > 1) {
    const poly g = power_of_x(piece, *phi);
    for (uint64_t j = 1; j < pieces; ++j) {
      starts[j] = starts[j - 1];
      jump(starts[j], g);
    }
  }
  std::vector<std::thread> workers;
  for (uint64_t j = 0; j < pieces; ++j) {
    const uint64_t lo = j * piece, hi = std::min(count, lo + piece);
    if (lo >= hi) break;
    workers.emplace_back(generate, out + lo, hi - lo, starts[j], shift);
  }
  for (auto& t : workers) t.join();
  return static_cast<int>(workers.size());
}
}
