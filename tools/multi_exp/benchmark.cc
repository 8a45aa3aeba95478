// Command-line clone of the reference's benchmark/multi_exp_pip and, compiled with
// -DBZ_TRIANGLE, benchmark/multi_exp_triangle: same arguments, same
// input recipe, same output lines, everything through the public C ABI (include/blitzar_api.h).
//
//   benchmark <curve> <n> <num_samples> <num_outputs> <element_nbytes> <verbose>
//     curve = curve25519 | bls12_381 | bls12-381 | bn254 | grumpkin
//
// Reference behaviour restated (benchmark/multi_exp_pip/benchmark.m.cc, .../multi_exp_triangle):
//   :84-112    generators: curve25519 -> compute_base_element(i); the Weierstrass curves ->
//              generate_random_element(fast_random_number_generator{i + 1, i + 2}) = the curve's
//              generator times 32 random bytes with bit 255 ignored
//              (sxt/curve_bng1/random/element_p2.h:33-46, operation/scalar_multiply.cc:36-101)
//   :117-131   exponents[byte + nbytes * output + nbytes * num_outputs * row] from std::mt19937{0}
//              through uniform_int_distribution<uint8_t>, drawn output-major
//   :180-206   one discarded run, then the mean of `num_samples` runs in whole milliseconds
//   triangle   output i has bit width 8 * nbytes and length min(++counter, n), counter starting at
//              max(n - num_outputs, 0)  (multi_exp_triangle/benchmark.m.cc:129-145)
// The reference links the library's internals; here a fixed-base handle is made with
// sxt_multiexp_handle_new and the runs are sxt_fixed_multiexponentiation /
// sxt_fixed_vlen_multiexponentiation.  The Weierstrass generators s_i * G come out of the library
// too: a handle over the single generator G evaluated with one 32-byte scalar per output.
// The backend is the GPU unless BLITZAR_BACKEND=cpu (the override sxt_init honours).
// Verbose output prints canonical forms like the reference's operator<< of the compressed /
// affine types; canonicalisation uses this library's bzamd_fold_encode (include/blitzar_amd.h).
//
//   g++ -O2 -std=c++17 -I include tools/multi_exp/benchmark.cc [-DBZ_TRIANGLE] \
//       -L blitzar_amd/lib -lblitzar_amd -Wl,-rpath,$PWD/blitzar_amd/lib -o multi_exp_pip
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "blitzar_amd.h"
#include "blitzar_api.h"

namespace {
using u64 = uint64_t;
using u128 = unsigned __int128;

// sxt/base/num/fast_random_number_generator.h:29-46 (xorshift128+)
struct fast_rng {
  u64 s0, s1;
  u64 operator()() {
    u64 x = s0;
    const u64 y = s1;
    s0 = y;
    x ^= x << 23;
    s1 = x ^ y ^ (x >> 17) ^ (y >> 26);
    return s1 + y;
  }
};

// little-endian multi-word helpers for the base fields (N = 4 or 6 words)
template <int N> bool geq(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
template <int N> void sub(u64* a, const u64* b) {
  u64 borrow = 0;
  for (int i = 0; i < N; ++i) {
    const u128 d = static_cast<u128>(a[i]) - b[i] - borrow;
    a[i] = static_cast<u64>(d);
    borrow = static_cast<u64>(d >> 64) & 1;
  }
}
// a = 2 a mod p, a < p < 2^(64 N - 1)
template <int N> void double_mod(u64* a, const u64* p) {
  for (int i = N - 1; i > 0; --i) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
  a[0] <<= 1;
  if (geq<N>(a, p)) sub<N>(a, p);
}
// x * 2^(64 N) mod p: the ABI's Montgomery form of a canonical x
template <int N> void to_montgomery(u64* x, const u64* p) {
  for (int i = 0; i < 64 * N; ++i) double_mod<N>(x, p);
}
// x / 2^(64 N) mod p by word-wise Montgomery reduction (p odd)
template <int N> void from_montgomery(u64* x, const u64* p) {
  u64 inv = 1; // -p^-1 mod 2^64 by Newton iteration
  for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
  inv = ~inv + 1;
  u64 t[2 * N + 1] = {};
  std::memcpy(t, x, 8 * N);
  for (int i = 0; i < N; ++i) {
    const u64 m = t[i] * inv;
    u128 carry = 0;
    for (int j = 0; j < N; ++j) {
      const u128 s = static_cast<u128>(m) * p[j] + t[i + j] + carry;
      t[i + j] = static_cast<u64>(s);
      carry = s >> 64;
    }
    for (int j = i + N; carry != 0 && j <= 2 * N; ++j) {
      const u128 s = static_cast<u128>(t[j]) + carry;
      t[j] = static_cast<u64>(s);
      carry = s >> 64;
    }
  }
  std::memcpy(x, t + N, 8 * N);
  if (t[2 * N] != 0 || geq<N>(x, p)) sub<N>(x, p);
}

struct curve_info {
  unsigned id;
  int words;              // 64-bit words of a base-field element
  const u64* p;           // modulus
  const u64* gx;          // canonical generator coordinates
  const u64* gy;
  size_t projective_size; // sxt_*_p2 / sxt_ristretto255
  const char* suffix;     // of the reference's field operator<<
};

const u64 kBn254P[4] = {0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029};
const u64 kBn254Gx[4] = {1, 0, 0, 0}, kBn254Gy[4] = {2, 0, 0, 0};
const u64 kGrumpkinP[4] = {0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029};
const u64 kGrumpkinGx[4] = {1, 0, 0, 0}; // (1, sqrt(-16)), sxt/curve_gk/constant/generator.h:34-64
const u64 kGrumpkinGy[4] = {0x833fc48d823f272c, 0x2d270d45f1181294, 0xcf135e7506a45d63, 0x0000000000000002};
const u64 kBlsP[6] = {0xb9feffffffffaaab, 0x1eabfffeb153ffff, 0x6730d2a0f6b0f624,
                      0x64774b84f38512bf, 0x4b1ba7b6434bacd7, 0x1a0111ea397fe69a};
const u64 kBlsGx[6] = {0xfb3af00adb22c6bb, 0x6c55e83ff97a1aef, 0xa14e3a3f171bac58,
                       0xc3688c4f9774b905, 0x2695638c4fa9ac0f, 0x17f1d3a73197d794};
const u64 kBlsGy[6] = {0x0caa232946c5e7e1, 0xd03cc744a2888ae4, 0x00db18cb2c04b3ed,
                       0xfcf5e095d5d00af6, 0xa09e30ed741d8ae4, 0x08b3f481e3aaa0f1};

// generators[i] = (32 random bytes of rng{i + 1, i + 2}, bit 255 cleared) * G, as sxt_*_p2
template <int N>
std::vector<uint8_t> weierstrass_generators(const curve_info& c, unsigned n) {
  // G in the ABI's Montgomery projective form {X, Y, Z = 1}
  std::vector<u64> g(3 * N, 0);
  std::memcpy(g.data(), c.gx, 8 * N);
  std::memcpy(g.data() + N, c.gy, 8 * N);
  g[2 * N] = 1;
  for (int k = 0; k < 3; ++k) to_montgomery<N>(g.data() + k * N, c.p);
  std::vector<uint8_t> scalars(static_cast<size_t>(n) * 32);
  for (unsigned i = 0; i < n; ++i) {
    fast_rng rng{i + 1, i + 2};
    for (int k = 0; k < 32; k += 8) {
      const u64 x = rng();
      std::memcpy(scalars.data() + static_cast<size_t>(i) * 32 + k, &x, 8);
    }
    scalars[static_cast<size_t>(i) * 32 + 31] &= 0x7f; // scalar_multiply255 skips bit 255
  }
  std::vector<uint8_t> out(static_cast<size_t>(n) * c.projective_size);
  sxt_multiexp_handle* one = sxt_multiexp_handle_new(c.id, g.data(), 1);
  sxt_fixed_multiexponentiation(out.data(), one, 32, n, 1, scalars.data());
  sxt_multiexp_handle_free(one);
  return out;
}

template <int N> void print_field(const u64* montgomery, const curve_info& c) {
  u64 x[N];
  std::memcpy(x, montgomery, 8 * N);
  from_montgomery<N>(x, c.p);
  // sxt/field25/type/element.cc:27-57: 0x, the most significant non-zero byte unpadded, the rest
  // two digits each, then the field's suffix
  uint8_t bytes[8 * N];
  std::memcpy(bytes, x, 8 * N);
  std::cout << "0x";
  int start = 8 * N - 1;
  while (start >= 0 && bytes[start] == 0) --start;
  if (start < 0) {
    std::cout << "0" << c.suffix;
    return;
  }
  const auto flags = std::cout.flags();
  std::cout << std::hex << static_cast<int>(bytes[start]);
  for (int i = start; i-- > 0;) {
    std::cout << std::hex << std::setw(2) << std::setfill('0') << static_cast<int>(bytes[i]);
  }
  std::cout << c.suffix;
  std::cout.flags(flags);
}

void print_results(const curve_info& c, const std::vector<uint8_t>& res, unsigned num_outputs) {
  const size_t encoded = c.id == SXT_CURVE_RISTRETTO255 ? 32 : (c.id == SXT_CURVE_BLS_381 ? 48 : 72);
  std::vector<uint8_t> canonical(encoded);
  for (unsigned k = 0; k < num_outputs; ++k) {
    bzamd_fold_encode(c.id, canonical.data(), res.data() + k * c.projective_size, 1, 1);
    std::cout << k << ": ";
    if (c.id == SXT_CURVE_RISTRETTO255 || c.id == SXT_CURVE_BLS_381) {
      // byte list; the bls12-381 printer of the reference ends with a comma
      // (sxt/curve_g1/type/compressed_element.cc:34-45)
      std::cout << "{";
      for (size_t i = 0; i < encoded; ++i) {
        std::cout << static_cast<int>(canonical[i]);
        if (c.id == SXT_CURVE_BLS_381 || i + 1 != encoded) std::cout << ",";
      }
      std::cout << "}\n";
    } else {
      u64 xy[8];
      std::memcpy(xy, canonical.data(), 64);
      std::cout << "{";
      print_field<4>(xy, c);
      std::cout << ", ";
      print_field<4>(xy + 4, c);
      std::cout << "}\n";
    }
  }
}
} // namespace

int main(int argc, char* argv[]) {
  if (argc != 7) {
    std::cout << "Usage: benchmark <curve> <n> <num_samples> <num_outputs> <element_nbytes> <verbose>\n";
    return -1;
  }
  const std::string curve = argv[1];
  const unsigned n = static_cast<unsigned>(std::strtoul(argv[2], nullptr, 10));
  const unsigned num_samples = static_cast<unsigned>(std::strtoul(argv[3], nullptr, 10));
  const unsigned num_outputs = static_cast<unsigned>(std::strtoul(argv[4], nullptr, 10));
  const unsigned element_num_bytes = static_cast<unsigned>(std::strtoul(argv[5], nullptr, 10));
  const bool verbose = std::string(argv[6]) != "0";
  if (n == 0 || num_samples == 0 || num_outputs == 0 || element_num_bytes == 0 ||
      element_num_bytes > 32) {
    std::cout << "invalid argument\n";
    return -1;
  }
  std::cout << "n = " << n << "\n";
  std::cout << "num_samples = " << num_samples << "\n";
  std::cout << "num_outputs = " << num_outputs << "\n";
  std::cout << "element_num_bytes = " << element_num_bytes << "\n";

  curve_info info{};
  if (curve == "curve25519") {
    info = {SXT_CURVE_RISTRETTO255, 0, nullptr, nullptr, nullptr, sizeof(sxt_ristretto255), ""};
  } else if (curve == "bls12_381" || curve == "bls12-381") {
    info = {SXT_CURVE_BLS_381, 6, kBlsP, kBlsGx, kBlsGy, sizeof(sxt_bls12_381_g1_p2), "_f12"};
  } else if (curve == "bn254") {
    info = {SXT_CURVE_BN_254, 4, kBn254P, kBn254Gx, kBn254Gy, sizeof(sxt_bn254_g1_p2), "_f25"};
  } else if (curve == "grumpkin") {
    info = {SXT_CURVE_GRUMPKIN, 4, kGrumpkinP, kGrumpkinGx, kGrumpkinGy, sizeof(sxt_grumpkin_p2),
            "_fgk"};
  } else {
    std::cout << "curve not supported\n";
    return 0;
  }
  std::cout << "running " << curve << " benchmark...\n";
  const sxt_config config{SXT_GPU_BACKEND, 0}; // BLITZAR_BACKEND=cpu overrides
  if (sxt_init(&config) != 0) return -1;

  std::vector<uint8_t> generators;
  if (info.id == SXT_CURVE_RISTRETTO255) {
    generators.resize(static_cast<size_t>(n) * sizeof(sxt_ristretto255));
    if (sxt_ristretto255_get_generators(reinterpret_cast<sxt_ristretto255*>(generators.data()), n,
                                        0) != 0) {
      return -1;
    }
  } else if (info.words == 4) {
    generators = weierstrass_generators<4>(info, n);
  } else {
    generators = weierstrass_generators<6>(info, n);
  }
  sxt_multiexp_handle* handle = sxt_multiexp_handle_new(info.id, generators.data(), n);

  std::vector<uint8_t> exponents(static_cast<size_t>(num_outputs) * n * element_num_bytes);
  {
    std::mt19937 rng{0};
    std::uniform_int_distribution<uint8_t> dist{0, UINT8_MAX};
    for (unsigned output = 0; output < num_outputs; ++output) {
      for (unsigned i = 0; i < n; ++i) {
        for (unsigned b = 0; b < element_num_bytes; ++b) {
          exponents[b + static_cast<size_t>(element_num_bytes) * output +
                    static_cast<size_t>(element_num_bytes) * num_outputs * i] = dist(rng);
        }
      }
    }
  }
  std::vector<uint8_t> res(static_cast<size_t>(num_outputs) * info.projective_size);
#ifdef BZ_TRIANGLE
  std::vector<unsigned> bit_widths(num_outputs, element_num_bytes * 8), lengths(num_outputs);
  unsigned counter = n > num_outputs ? n - num_outputs : 0;
  for (unsigned i = 0; i < num_outputs; ++i) lengths[i] = std::min(++counter, n);
  auto run = [&] {
    sxt_fixed_vlen_multiexponentiation(res.data(), handle, bit_widths.data(), lengths.data(),
                                       num_outputs, exponents.data());
  };
#else
  auto run = [&] {
    sxt_fixed_multiexponentiation(res.data(), handle, element_num_bytes, num_outputs, n,
                                  exponents.data());
  };
#endif
  run(); // discard initial run
  double times = 0;
  for (unsigned i = 0; i < num_samples; ++i) {
    const auto t1 = std::chrono::steady_clock::now();
    run();
    const auto t2 = std::chrono::steady_clock::now();
    times += std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() / 1e3;
  }
  if (verbose) print_results(info, res, num_outputs);
  std::cout << "compute duration (s): " << times / num_samples << "\n";
  sxt_multiexp_handle_free(handle);
  return 0;
}
