// Inner-product argument prover / verifier (proof/inner_product.h).
//
// Protocol (reference: sxt/proof/inner_product/proof_computation.cc:54-155): with np = 2^k >= n,
// generators g_0 .. g_{np-1}, Q = g_np, vectors a, b padded with zeros to np; per round
//   L = <a_lo, g_hi> + <a_lo, b_hi> Q,   R = <a_hi, g_lo> + <a_hi, b_lo> Q,
//   x = challenge(L, R),
//   a' = x a_lo + x^-1 a_hi,  b' = x^-1 b_lo + x b_hi,  g' = x^-1 g_lo + x g_hi,
// until one element is left.  Proof bytes (compressed L, R; the last a) are canonical encodings
// of group elements / scalars, so any correct evaluation order reproduces the reference's bytes.
//
// Work split (GPU backend): the two MSMs of a round run on the MSM engine with the folded
// generators resident in HBM (the reference's gpu_driver re-uploads nothing either,
// gpu_driver.cc:60-130); scalar folds, inner products and the generator fold are kernels below;
// the transcript, the challenge inversions and the two c Q products are host work.  The host
// backend runs the same round loop on host loops.
#include "blitzar_amd/csrc/proof/inner_product.h"

#include <algorithm>
#include <memory>
#include <vector>

#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/proof/scalar25.h"
#include "blitzar_amd/csrc/proof/transcript.h"

namespace bz::proof {
namespace {
using s25::scalar;
constexpr u32 kScalarBits = 253; // s25cn::max_bits_v

// the joint bit pattern of (m_low, m_high), least significant first: digit = bit(m_low) +
// 2 bit(m_high); trailing zero digits dropped (generator_fold.cc:32-59)
struct fold_digits {
  u8 d[256];
  u32 count;
};

fold_digits decompose_fold(const u8 m_low[32], const u8 m_high[32]) {
  fold_digits r{};
  for (u32 bit = 0; bit < kScalarBits; ++bit) {
    const u32 lo = (m_low[bit >> 3] >> (bit & 7)) & 1, hi = (m_high[bit >> 3] >> (bit & 7)) & 1;
    r.d[bit] = static_cast<u8>(lo + 2 * hi);
  }
  r.count = kScalarBits;
  while (r.count > 0 && r.d[r.count - 1] == 0) --r.count;
  return r;
}

// m_low g_low + m_high g_high by one shared double-and-add over the joint digits
// (generator_fold.cc:64-90); `term(k)` yields g_low, g_high, g_low + g_high as cached addends
template <class Term> BZ_HD ed29_point fold_point(const fold_digits& digits, Term&& term) {
  ed29_point acc = ed29::identity();
  for (u32 bit = digits.count; bit-- > 0;) {
    const u32 d = digits.d[bit];
    // T is only needed by a following addition and by the caller (the last step)
    if (bit + 1 != digits.count) acc = ed29::dbl(acc, d != 0 || bit == 0);
    if (d != 0) acc = ed29::add_cached(acc, term(d - 1), false);
  }
  return acc;
}

// k * p on the host (253-bit double-and-add on the ABI-form arithmetic)
ed_point scalar_multiply(const ed_point& p, const u8 k[32]) {
  ed_point acc = ed::identity();
  bool started = false;
  for (int bit = 255; bit >= 0; --bit) {
    if (started) acc = ed::dbl(acc);
    if ((k[bit >> 3] >> (bit & 7)) & 1) {
      acc = started ? ed::add(acc, p) : p;
      started = true;
    }
  }
  return acc;
}

//--------------------------------------------------------------------------------------------------
// device kernels
//--------------------------------------------------------------------------------------------------
// out[i] = m_low x[i] + m_high x[mid + i] (canonical), i < mid; x has `len` entries, the missing
// high ones count as zero (fold.cc:30-45).  m_* in Montgomery form, x plain: the products are plain.
// In place (out == x) is safe: entry i is only read by its own lane, entries >= mid are not written.
__global__ void __launch_bounds__(256)
    k_fold_scalars(u64* out, const u64* x, s25::fe m_low, s25::fe m_high, u32 mid,
                   u32 len) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mid) return;
  s25::fe r = s25::F::mul(m_low, s25::load_words(x + 4 * static_cast<u64>(i)));
  if (mid + i < len) {
    r = s25::add(r, s25::F::mul(m_high, s25::load_words(x + 4 * static_cast<u64>(mid + i))));
  }
  u64 w[4];
  s25::store_words(w, r);
  for (int k = 0; k < 4; ++k) out[4 * static_cast<u64>(i) + k] = w[k];
}

// partials[block] = sum over the block's share of a[i] b[i] / R (plain inputs; the host adds the
// partials and multiplies by R^2): grid-stride products, LDS tree
__global__ void __launch_bounds__(256)
    k_inner_product(s25::fe* __restrict__ partials, const u64* __restrict__ a,
                    const u64* __restrict__ b, u32 count) {
  __shared__ s25::fe tree[256];
  s25::fe acc = s25::F::zero();
  for (u64 i = blockIdx.x * 256u + threadIdx.x; i < count; i += static_cast<u64>(gridDim.x) * 256u) {
    acc = s25::add(acc, s25::F::mul(s25::load_words(a + 4 * i), s25::load_words(b + 4 * i)));
  }
  tree[threadIdx.x] = acc;
  __syncthreads();
  for (u32 stride = 128; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) {
      tree[threadIdx.x] = s25::add(tree[threadIdx.x], tree[threadIdx.x + stride]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = tree[0];
}

// terms[k * mid + i], k = 0, 1, 2: g_i, g_{mid + i} and their sum as packed cached addends
__global__ void __launch_bounds__(256)
    k_fold_terms(ed29_cached_packed* __restrict__ terms, const ed_point* __restrict__ g, u32 mid) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mid) return;
  const ed29_point lo = ed29::from_ed(g[i]), hi = ed29::from_ed(g[mid + i]);
  terms[i] = ed29::pack(ed29::to_cached(lo));
  terms[mid + i] = ed29::pack(ed29::to_cached(hi));
  terms[2 * static_cast<u64>(mid) + i] = ed29::pack(ed29::to_cached(ed29::add(lo, hi)));
}

// out[i] = m_low g_i + m_high g_{mid + i}: every lane walks the SAME digit sequence (the digits
// are a kernel argument: uniform control flow), gathering its own three terms
__global__ void __launch_bounds__(256)
    k_fold_generators(ed_point* __restrict__ out, const ed29_cached_packed* __restrict__ terms,
                      fold_digits digits, u32 mid) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mid) return;
  const ed29_point r = fold_point(digits, [&](u32 k) {
    return ed29::unpack(terms[static_cast<u64>(k) * mid + i]);
  });
  out[i] = ed29::to_ed(r);
}

//--------------------------------------------------------------------------------------------------
// the computational backends of the round loop (reference: prfip::driver, driver.h:35-95)
//--------------------------------------------------------------------------------------------------
class fold_backend {
public:
  virtual ~fold_backend() = default;
  // c_l = <a_lo, b_hi>, c_r = <a_hi, b_lo> (canonical bytes); l_p = <a_lo, g_hi>, r_p = <a_hi, g_lo>
  virtual void commit_to_fold(u8 c_l[32], u8 c_r[32], ed_point& l_p, ed_point& r_p) = 0;
  // a' = x a_lo + x^-1 a_hi; unless one element is left: b' = x^-1 b_lo + x b_hi,
  // g' = x^-1 g_lo + x g_hi
  virtual void fold(const scalar& x, const scalar& x_inv) = 0;
  virtual void first_a(u8 out[32]) = 0;
};

// current vector shapes, shared by both backends: g has `size` entries, a and b have
// min(size, their original length) -- only round 0 can be ragged
struct fold_shape {
  u64 size, a_len, b_len;
  u64 mid() const { return size / 2; }
  void advance() {
    size = mid();
    a_len = size;
    b_len = size;
  }
};

class host_fold_backend final : public fold_backend {
public:
  host_fold_backend(api_state& st, u64 n, u64 np, u64 offset, const u8* a, const u8* b)
      : st_{st}, shape_{np, n, n}, a_(a, a + 32 * n), b_(b, b + 32 * n), g_(np) {
    host_builtin_generators_unlocked(st, g_.data(), np, offset);
  }

  void commit_to_fold(u8 c_l[32], u8 c_r[32], ed_point& l_p, ed_point& r_p) override {
    const u64 mid = shape_.mid();
    inner_product(c_l, a_.data(), b_.data() + 32 * mid, std::min(mid, shape_.b_len - mid));
    inner_product(c_r, a_.data() + 32 * mid, b_.data(), std::min(shape_.a_len - mid, mid));
    l_p = msm(a_.data(), mid, g_.data() + mid);
    r_p = msm(a_.data() + 32 * mid, shape_.a_len - mid, g_.data());
  }

  void fold(const scalar& x, const scalar& x_inv) override {
    const u64 mid = shape_.mid();
    fold_scalars(a_, x, x_inv, mid, shape_.a_len);
    if (mid > 1) {
      fold_scalars(b_, x_inv, x, mid, shape_.b_len);
      u8 lo[32], hi[32];
      x_inv.to_bytes(lo);
      x.to_bytes(hi);
      const fold_digits digits = decompose_fold(lo, hi);
      for (u64 i = 0; i < mid; ++i) {
        const ed29_point g_lo = ed29::from_ed(g_[i]), g_hi = ed29::from_ed(g_[mid + i]);
        const ed29_cached terms[3] = {ed29::to_cached(g_lo), ed29::to_cached(g_hi),
                                      ed29::to_cached(ed29::add(g_lo, g_hi))};
        g_[i] = ed29::to_ed(fold_point(digits, [&](u32 k) { return terms[k]; }));
      }
    }
    shape_.advance();
  }

  void first_a(u8 out[32]) override { std::memcpy(out, a_.data(), 32); }

private:
  api_state& st_;
  fold_shape shape_;
  std::vector<u8> a_, b_;
  std::vector<ed_point> g_;

  static void inner_product(u8 out[32], const u8* a, const u8* b, u64 count) {
    s25::fe acc = s25::F::zero();
    for (u64 i = 0; i < count; ++i) {
      acc = s25::add(acc, s25::F::mul(s25::load(a + 32 * i), s25::load(b + 32 * i)));
    }
    s25::store(out, s25::F::mul(acc, s25::r2()));
  }
  static void fold_scalars(std::vector<u8>& x, const scalar& m_low, const scalar& m_high, u64 mid,
                           u64 len) {
    for (u64 i = 0; i < mid; ++i) {
      s25::fe r = s25::F::mul(m_low.m, s25::load(x.data() + 32 * i));
      if (mid + i < len) {
        r = s25::add(r, s25::F::mul(m_high.m, s25::load(x.data() + 32 * (mid + i))));
      }
      s25::store(x.data() + 32 * i, r);
    }
  }
  ed_point msm(const u8* scalars, u64 count, const ed_point* generators) {
    ed_point r = ed::identity();
    if (count == 0) return r;
    const std::vector<host_column> cols{byte_column(scalars, count, 32, false)};
    curve25519_vtable().msm_host(reinterpret_cast<u8*>(&r), sizeof(ed_point), true, cols,
                                 generators, false, count);
    return r;
  }
};

class device_fold_backend final : public fold_backend {
public:
  device_fold_backend(api_state& st, u64 n, u64 np, u64 offset, const u8* a, const u8* b)
      : ds_{st.primary()}, shape_{np, n, n} {
    ds_.activate();
    const u64 mid = np / 2;
    ds_.io.reset(2 * device_arena::padded(32 * n) + device_arena::padded(sizeof(ed_point) * np) +
                     device_arena::padded(sizeof(ed29_cached_packed) * 3 * mid) +
                     2 * device_arena::padded(sizeof(s25::fe) * kPartialBlocks) +
                     device_arena::padded(2 * sizeof(ed_point)) + 4096,
                 ds_.stream);
    d_a_ = ds_.io.take<u64>(4 * n);
    d_b_ = ds_.io.take<u64>(4 * n);
    d_g_ = ds_.io.take<ed_point>(np);
    d_terms_ = ds_.io.take<ed29_cached_packed>(3 * mid);
    d_partials_ = ds_.io.take<s25::fe>(2 * kPartialBlocks);
    d_msm_ = ds_.io.take<ed_point>(2);
    BZ_HIP_CHECK(hipMemcpyAsync(d_a_, a, 32 * n, hipMemcpyHostToDevice, ds_.stream));
    BZ_HIP_CHECK(hipMemcpyAsync(d_b_, b, 32 * n, hipMemcpyHostToDevice, ds_.stream));
    builtin_generators_enqueue(d_g_, offset, np, ds_.stream);
    g_kernel_launches += 1;
  }

  void commit_to_fold(u8 c_l[32], u8 c_r[32], ed_point& l_p, ed_point& r_p) override {
    const u64 mid = shape_.mid();
    const u64 n_l = std::min(mid, shape_.b_len - mid), n_r = std::min(shape_.a_len - mid, mid);
    const u32 blocks_l = blocks_for(n_l), blocks_r = blocks_for(n_r);
    if (n_l > 0) {
      hipLaunchKernelGGL(k_inner_product, dim3(blocks_l), dim3(256), 0, ds_.stream, d_partials_,
                         d_a_, d_b_ + 4 * mid, static_cast<u32>(n_l));
    }
    if (n_r > 0) {
      hipLaunchKernelGGL(k_inner_product, dim3(blocks_r), dim3(256), 0, ds_.stream,
                         d_partials_ + kPartialBlocks, d_a_ + 4 * mid, d_b_, static_cast<u32>(n_r));
    }
    BZ_HIP_CHECK(hipGetLastError());
    g_kernel_launches += 2;
    // the two MSMs of the round on the engine: generators and scalars already resident
    const curve_vtable& vt = curve25519_vtable();
    const std::vector<host_column> col_l{
        byte_column(reinterpret_cast<const u8*>(d_a_), mid, 32, false)};
    vt.msm(*ds_.ctx, reinterpret_cast<u8*>(d_msm_), sizeof(ed_point), true, col_l, nullptr,
           d_g_ + mid, ds_.stream);
    const std::vector<host_column> col_r{
        byte_column(reinterpret_cast<const u8*>(d_a_ + 4 * mid), shape_.a_len - mid, 32, false)};
    vt.msm(*ds_.ctx, reinterpret_cast<u8*>(d_msm_ + 1), sizeof(ed_point), true, col_r, nullptr,
           d_g_, ds_.stream);
    std::vector<s25::fe> partials(2 * kPartialBlocks);
    ed_point results[2];
    BZ_HIP_CHECK(hipMemcpyAsync(partials.data(), d_partials_, sizeof(s25::fe) * 2 * kPartialBlocks,
                                hipMemcpyDeviceToHost, ds_.stream));
    BZ_HIP_CHECK(hipMemcpyAsync(results, d_msm_, sizeof(results), hipMemcpyDeviceToHost,
                                ds_.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(ds_.stream));
    finish_inner_product(c_l, partials.data(), n_l > 0 ? blocks_l : 0);
    finish_inner_product(c_r, partials.data() + kPartialBlocks, n_r > 0 ? blocks_r : 0);
    l_p = results[0];
    r_p = results[1];
  }

  void fold(const scalar& x, const scalar& x_inv) override {
    const u32 mid = static_cast<u32>(shape_.mid());
    const u32 blocks = (mid + 255) / 256;
    hipLaunchKernelGGL(k_fold_scalars, dim3(blocks), dim3(256), 0, ds_.stream, d_a_, d_a_, x.m,
                       x_inv.m, mid, static_cast<u32>(shape_.a_len));
    g_kernel_launches += 1;
    if (mid > 1) {
      hipLaunchKernelGGL(k_fold_scalars, dim3(blocks), dim3(256), 0, ds_.stream, d_b_, d_b_,
                         x_inv.m, x.m, mid, static_cast<u32>(shape_.b_len));
      u8 lo[32], hi[32];
      x_inv.to_bytes(lo);
      x.to_bytes(hi);
      const fold_digits digits = decompose_fold(lo, hi);
      hipLaunchKernelGGL(k_fold_terms, dim3(blocks), dim3(256), 0, ds_.stream, d_terms_, d_g_, mid);
      hipLaunchKernelGGL(k_fold_generators, dim3(blocks), dim3(256), 0, ds_.stream, d_g_, d_terms_,
                         digits, mid);
      g_kernel_launches += 3;
    }
    BZ_HIP_CHECK(hipGetLastError());
    shape_.advance();
  }

  void first_a(u8 out[32]) override {
    BZ_HIP_CHECK(hipMemcpyAsync(out, d_a_, 32, hipMemcpyDeviceToHost, ds_.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(ds_.stream));
  }

private:
  static constexpr u32 kPartialBlocks = 256;
  device_state& ds_;
  fold_shape shape_;
  u64* d_a_ = nullptr;
  u64* d_b_ = nullptr;
  ed_point* d_g_ = nullptr;
  ed29_cached_packed* d_terms_ = nullptr;
  s25::fe* d_partials_ = nullptr;
  ed_point* d_msm_ = nullptr;

  static u32 blocks_for(u64 count) {
    return static_cast<u32>(std::min<u64>(kPartialBlocks, (count + 255) / 256));
  }
  static void finish_inner_product(u8 out[32], const s25::fe* partials, u32 blocks) {
    s25::fe acc = s25::F::zero();
    for (u32 k = 0; k < blocks; ++k) acc = s25::add(acc, partials[k]);
    s25::store(out, s25::F::mul(acc, s25::r2()));
  }
};

std::unique_ptr<fold_backend> make_backend(api_state& st, u64 n, u64 np, u64 offset, const u8* a,
                                           const u8* b) {
  if (st.backend == 2) return std::make_unique<device_fold_backend>(st, n, np, offset, a, b);
  return std::make_unique<host_fold_backend>(st, n, np, offset, a, b);
}

u64 ceil_log2(u64 n) {
  u64 k = 0;
  while ((u64{1} << k) < n) ++k;
  return k;
}

// proof_computation.cc:36-52
void init_transcript(transcript& t, u64 n) {
  t.set_domain("inner product proof v1");
  t.append_u64("n", n);
}
scalar round_challenge(transcript& t, const u8* l_value, const u8* r_value) {
  t.append_message("L", l_value, 32);
  t.append_message("R", r_value, 32);
  u8 x[32];
  t.challenge_bytes(x, 32, "x");
  return scalar::from_bytes(x); // == s25o::reduce32: every later use is modulo l
}
} // namespace

void prove_inner_product(api_state& st, u8* l_vector, u8* r_vector, u8* ap_value,
                         void* transcript_bytes, u64 n, u64 generators_offset, const u8* a_vector,
                         const u8* b_vector) {
  transcript t{transcript_bytes};
  init_transcript(t, n);
  if (n == 1) {
    std::memcpy(ap_value, a_vector, 32); // verbatim, as proof_computation.cc:83-86
    return;
  }
  const u64 np = u64{1} << ceil_log2(n);
  ed_point q;
  host_builtin_generators_unlocked(st, &q, 1, generators_offset + np);
  std::unique_ptr<fold_backend> backend =
      make_backend(st, n, np, generators_offset, a_vector, b_vector);
  u64 round = 0;
  for (u64 size = np; size > 1; size /= 2, ++round) {
    u8 c_l[32], c_r[32];
    ed_point l_p, r_p;
    backend->commit_to_fold(c_l, c_r, l_p, r_p);
    l_p = ed::add(l_p, scalar_multiply(q, c_l));
    r_p = ed::add(r_p, scalar_multiply(q, c_r));
    u8* l_value = l_vector + 32 * round;
    u8* r_value = r_vector + 32 * round;
    ristretto::encode(l_value, l_p);
    ristretto::encode(r_value, r_p);
    const scalar x = round_challenge(t, l_value, r_value);
    backend->fold(x, x.inverse());
  }
  backend->first_a(ap_value);
}

bool verify_inner_product(api_state& st, void* transcript_bytes, u64 n, u64 generators_offset,
                          const u8* b_vector, const u8* product, const void* a_commit,
                          const u8* l_vector, const u8* r_vector, const u8* ap_value) {
  const u64 rounds = ceil_log2(n), np = u64{1} << rounds;
  transcript t{transcript_bytes};
  init_transcript(t, n);
  std::vector<scalar> x(rounds);
  for (u64 i = 0; i < rounds; ++i) x[i] = round_challenge(t, l_vector + 32 * i, r_vector + 32 * i);

  // exponents of [Q, g_0 .. g_{np-1}, L_0 .., R_0 ..] (verification_computation.cc:30-121)
  const u64 count = 1 + np + 2 * rounds;
  std::vector<u8> exponents(32 * count);
  const scalar ap = scalar::from_bytes(ap_value);
  std::vector<scalar> g_exponents(np);
  if (n == 1) {
    (scalar::from_bytes(b_vector) * ap).to_bytes(exponents.data());
    g_exponents[0] = ap;
  } else {
    std::vector<scalar> x_sq(rounds);
    scalar all_inv = x[0].inverse();
    x_sq[0] = x[0] * x[0];
    (-(all_inv * all_inv)).to_bytes(exponents.data() + 32 * (1 + np + rounds));
    for (u64 i = 1; i < rounds; ++i) {
      const scalar xi_inv = x[i].inverse();
      all_inv = all_inv * xi_inv;
      x_sq[i] = x[i] * x[i];
      (-(xi_inv * xi_inv)).to_bytes(exponents.data() + 32 * (1 + np + rounds + i));
    }
    g_exponents[0] = all_inv * ap;
    u64 a = 1, b = 2, next = rounds;
    while (a != np) {
      const scalar multiplier = x_sq[--next];
      for (u64 i = a; i < b; ++i) g_exponents[i] = multiplier * g_exponents[i - a];
      a = b;
      b = 2 * a;
    }
    s25::fe acc = s25::F::zero(); // <g_exponents, b> over the n entries of b
    for (u64 i = 0; i < n; ++i) {
      acc = s25::add(acc, s25::F::mul(g_exponents[i].m, s25::load(b_vector + 32 * i)));
    }
    s25::store(exponents.data(), acc);
    for (u64 i = 0; i < rounds; ++i) (-x_sq[i]).to_bytes(exponents.data() + 32 * (1 + np + i));
  }
  for (u64 i = 0; i < np; ++i) g_exponents[i].to_bytes(exponents.data() + 32 * (1 + i));

  std::vector<ed_point> generators(count);
  host_builtin_generators_unlocked(st, generators.data(), 1, generators_offset + np);
  host_builtin_generators_unlocked(st, generators.data() + 1, np, generators_offset);
  for (u64 i = 0; i < rounds; ++i) {
    // an invalid encoding cannot be part of a valid proof
    if (!ristretto::decode(generators[1 + np + i], l_vector + 32 * i)) return false;
    if (!ristretto::decode(generators[1 + np + rounds + i], r_vector + 32 * i)) return false;
  }
  u8 expected[32], commit[32];
  commit_column_unlocked(st, expected, exponents.data(), count, generators.data());
  // product * Q + a_commit (proof_computation.cc:146-153)
  ed_point a_point;
  std::memcpy(&a_point, a_commit, sizeof(a_point));
  ristretto::encode(commit, ed::add(scalar_multiply(generators[0], product), a_point));
  return std::memcmp(commit, expected, 32) == 0;
}
} // namespace bz::proof
