// Sumcheck prover (sxt_prove_sumcheck; reference: cbindings/sumcheck.cc, sxt/cbindings/backend/
// cpu_backend.cc:73-112, sxt/proof/sumcheck/{proof_computation,cpu_driver,polynomial_utility}.h).
//
// The polynomial is  sum_p mult_p * prod_{j in terms_p} f_j(X_1 .. X_r)  over multilinear
// extensions f_j given by their n evaluations (column-major n x num_mles).  Round t fixes the top
// variable: with mid = 2^(r-1-t), every row i < mid contributes  mult_p * prod_j (a_j + b_j X),
// a_j = f_j[i], b_j = f_j[mid + i] - a_j (rows without a partner: b_j = -a_j), to the round
// polynomial; the caller's transcript callback turns the polynomial into the challenge r; the
// tables fold to (1 - r) f[i] + r f[mid + i].  Outputs are field elements in the caller's
// representation, canonical, hence identical to the reference's bytes.
//
// Fields (blitzar_api.h SXT_FIELD_*): 0 = curve25519 scalar field (32 little-endian bytes,
// proof/scalar25.h), 1 = Grumpkin base field (4 x 64-bit Montgomery limbs, field/mont29.h
// grumpkin_fq29).  Both compute on the 9 x 29-bit Montgomery representation.
//
// GPU backend: the tables live in HBM in engine form; k_sumcheck_round reduces every row's
// contribution to per-workgroup partial polynomials, k_sumcheck_fold folds.  The host backend runs
// the same loops on the host.
#include "blitzar_amd/csrc/proof/sumcheck.h"

#include <algorithm>
#include <cstring>
#include <vector>

#include "blitzar_amd/csrc/field/mont29.h"
#include "blitzar_amd/csrc/proof/scalar25.h"

namespace bz::proof {
namespace {
constexpr u32 kMaxDegree = 8; // round polynomials of degree <= 8 (9 coefficients in registers)
constexpr u32 kRoundThreads = 128;
constexpr u32 kRoundBlocks = 512;

// caller representation <-> engine representation (Montgomery, normalised, V < 4)
struct scalar25519_elements {
  using F = scalar25_field;
  static constexpr u32 element_bytes = 32, product_stride = 36;
  BZ_HD static F::fe load(const u8* p) { return s25::to_mont(s25::load(p)); }
  BZ_HD static void store(u8* p, const F::fe& v) { s25::store(p, s25::from_mont(v)); }
};
struct grumpkin_elements {
  using F = grumpkin_fq29;
  static constexpr u32 element_bytes = 32, product_stride = 40;
  BZ_HD static F::fe load(const u8* p) {
    u64 w[4];
    std::memcpy(w, p, 32);
    return F::from_mont64(w);
  }
  BZ_HD static void store(u8* p, const F::fe& v) {
    u64 w[4];
    F::to_mont64(w, v);
    std::memcpy(p, w, 32);
  }
};

template <class F> BZ_HD typename F::fe fadd(const typename F::fe& a, const typename F::fe& b) {
  return F::reduce(F::norm(F::add(a, b)));
}
template <class F> BZ_HD typename F::fe fsub(const typename F::fe& a, const typename F::fe& b) {
  return F::reduce(F::norm(F::template sub<8>(a, b)));
}
template <class F> BZ_HD typename F::fe fneg(const typename F::fe& a) {
  return F::reduce(F::norm(F::template neg<8>(a)));
}

// product p (engine form): multiplier, terms [first_term, first_term + num_terms)
template <class F> struct product_desc {
  typename F::fe multiplier;
  u32 first_term, num_terms;
};

// poly[0 .. degree] += sum_products mult * prod_j (a_j + b_j X) for row i of tables of `n` rows
// (polynomial_utility.h:64-137 expand_products / partial_expand_products; cpu_driver.h:75-102)
template <class F>
BZ_HD void accumulate_row(typename F::fe* poly, const typename F::fe* mles, u64 n, u64 mid, u64 i,
                          const product_desc<F>* products, u32 num_products, const u32* terms) {
  using fe = typename F::fe;
  const bool paired = mid + i < n;
  for (u32 pi = 0; pi < num_products; ++pi) {
    const product_desc<F>& pd = products[pi];
    fe p[kMaxDegree + 1];
    for (u32 t = 0; t < pd.num_terms; ++t) {
      const fe* column = mles + static_cast<u64>(terms[pd.first_term + t]) * n;
      const fe a = column[i];
      const fe b = paired ? fsub<F>(column[mid + i], a) : fneg<F>(a);
      if (t == 0) {
        p[0] = a;
        p[1] = b;
        continue;
      }
      // p <- p * (a + b X)
      fe previous = p[0];
      p[0] = F::mul(previous, a);
      for (u32 k = 1; k <= t; ++k) {
        const fe current = p[k];
        p[k] = fadd<F>(F::mul(current, a), F::mul(previous, b));
        previous = current;
      }
      p[t + 1] = F::mul(previous, b);
    }
    for (u32 k = 0; k <= pd.num_terms; ++k) {
      poly[k] = fadd<F>(poly[k], F::mul(pd.multiplier, p[k]));
    }
  }
}

//--------------------------------------------------------------------------------------------------
// device kernels
//--------------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256)
    k_sumcheck_load(typename E::F::fe* __restrict__ out, const u8* __restrict__ elements, u64 count) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < count) out[i] = E::load(elements + E::element_bytes * i);
}

// partials[block][k] = the block's share of coefficient k of the round polynomial
template <class F>
__global__ void __launch_bounds__(kRoundThreads)
    k_sumcheck_round(typename F::fe* __restrict__ partials, const typename F::fe* __restrict__ mles,
                     u64 n, u64 mid, const product_desc<F>* __restrict__ products, u32 num_products,
                     const u32* __restrict__ terms, u32 degree) {
  using fe = typename F::fe;
  __shared__ fe tree[kRoundThreads];
  fe poly[kMaxDegree + 1];
  for (u32 k = 0; k <= kMaxDegree; ++k) poly[k] = F::zero();
  for (u64 i = static_cast<u64>(blockIdx.x) * kRoundThreads + threadIdx.x; i < mid;
       i += static_cast<u64>(gridDim.x) * kRoundThreads) {
    accumulate_row<F>(poly, mles, n, mid, i, products, num_products, terms);
  }
  for (u32 k = 0; k <= degree; ++k) {
    tree[threadIdx.x] = poly[k];
    __syncthreads();
    for (u32 stride = kRoundThreads / 2; stride > 0; stride >>= 1) {
      if (threadIdx.x < stride) {
        tree[threadIdx.x] = fadd<F>(tree[threadIdx.x], tree[threadIdx.x + stride]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) partials[static_cast<u64>(blockIdx.x) * (kMaxDegree + 1) + k] = tree[0];
    __syncthreads();
  }
}

// out[m * mid + i] = (1 - r) in[m * n + i] + r in[m * n + mid + i]  (cpu_driver.h:106-143)
template <class F>
__global__ void __launch_bounds__(256)
    k_sumcheck_fold(typename F::fe* __restrict__ out, const typename F::fe* __restrict__ in, u64 n,
                    u64 mid, u32 num_mles, typename F::fe r, typename F::fe one_minus_r) {
  const u64 id = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= mid * num_mles) return;
  const u64 m = id / mid, i = id % mid;
  typename F::fe v = F::mul(in[m * n + i], one_minus_r);
  if (mid + i < n) v = fadd<F>(v, F::mul(r, in[m * n + mid + i]));
  out[id] = v;
}

//--------------------------------------------------------------------------------------------------
template <class E>
void prove(api_state& st, u8* polynomials, u8* evaluation_point, const sumcheck_inputs& d,
           void* callback, void* context, api_state::device_lease* lease) {
  using F = typename E::F;
  using fe = typename F::fe;
  using callback_t = void (*)(void* r, void* context, const void* polynomial, unsigned length);
  const u32 degree = d.round_degree;
  const u32 length = degree + 1;
  u64 n = d.n;
  u32 num_variables = 0;
  while ((u64{1} << num_variables) < n) ++num_variables;
  if (num_variables == 0) num_variables = 1;

  // products in engine form
  std::vector<product_desc<F>> products(d.num_products);
  u32 first = 0;
  for (u32 p = 0; p < d.num_products; ++p) {
    const u8* entry = static_cast<const u8*>(d.product_table) + static_cast<size_t>(E::product_stride) * p;
    u32 num_terms;
    std::memcpy(&num_terms, entry + E::element_bytes, sizeof(num_terms));
    BZ_RELEASE_ASSERT(num_terms >= 1 && num_terms <= degree,
                      "a sumcheck product must have between 1 and round_degree terms");
    products[p] = product_desc<F>{E::load(entry), first, num_terms};
    first += num_terms;
  }
  BZ_RELEASE_ASSERT(first == d.num_product_terms, "num_product_terms does not match the product table");
  for (u32 t = 0; t < d.num_product_terms; ++t) {
    BZ_RELEASE_ASSERT(d.product_terms[t] < d.num_mles, "product term refers to a missing MLE");
  }

  const bool on_device = st.backend == 2;
  const u64 total = n * d.num_mles;
  std::vector<fe> h_mles, h_next;
  fe* d_mles = nullptr;
  fe* d_next = nullptr;
  fe* d_partials = nullptr;
  product_desc<F>* d_products = nullptr;
  u32* d_terms = nullptr;
  device_state* ds = nullptr;
  device_arena own; // not the device's staging arena: the lease is given up around the callback
  std::vector<fe> partials(static_cast<size_t>(kRoundBlocks) * (kMaxDegree + 1));
  if (on_device) {
    ds = &st.primary();
    ds->activate();
    const u64 half = (u64{1} << (num_variables - 1)) * d.num_mles;
    own.reset(device_arena::padded(static_cast<size_t>(E::element_bytes) * total) +
                     device_arena::padded(sizeof(fe) * total) + device_arena::padded(sizeof(fe) * half) +
                     device_arena::padded(sizeof(fe) * partials.size()) +
                     device_arena::padded(sizeof(product_desc<F>) * products.size()) +
                     device_arena::padded(sizeof(u32) * d.num_product_terms) + 4096,
                 ds->stream);
    u8* d_raw = own.take<u8>(static_cast<size_t>(E::element_bytes) * total);
    d_mles = own.take<fe>(total);
    d_next = own.take<fe>(half);
    d_partials = own.take<fe>(partials.size());
    d_products = own.take<product_desc<F>>(products.size());
    d_terms = own.take<u32>(d.num_product_terms);
    BZ_HIP_CHECK(hipMemcpyAsync(d_raw, d.mles, static_cast<size_t>(E::element_bytes) * total,
                                hipMemcpyHostToDevice, ds->stream));
    BZ_HIP_CHECK(hipMemcpyAsync(d_products, products.data(), sizeof(product_desc<F>) * products.size(),
                                hipMemcpyHostToDevice, ds->stream));
    BZ_HIP_CHECK(hipMemcpyAsync(d_terms, d.product_terms, sizeof(u32) * d.num_product_terms,
                                hipMemcpyHostToDevice, ds->stream));
    hipLaunchKernelGGL((k_sumcheck_load<E>), dim3(ceil_div_u32(total, 256)), dim3(256), 0,
                       ds->stream, d_mles, d_raw, total);
    BZ_HIP_CHECK(hipGetLastError());
    g_kernel_launches += 1;
  } else {
    h_mles.resize(total);
    const u8* raw = static_cast<const u8*>(d.mles);
    for (u64 i = 0; i < total; ++i) h_mles[i] = E::load(raw + E::element_bytes * i);
  }

  for (u32 round = 0; round < num_variables; ++round) {
    const u64 mid = u64{1} << (num_variables - 1 - round);
    std::vector<fe> poly(length, F::zero());
    if (on_device) {
      const u32 blocks = static_cast<u32>(std::min<u64>(kRoundBlocks, (mid + kRoundThreads - 1) / kRoundThreads));
      hipLaunchKernelGGL((k_sumcheck_round<F>), dim3(blocks), dim3(kRoundThreads), 0, ds->stream,
                         d_partials, d_mles, n, mid, d_products, d.num_products, d_terms, degree);
      BZ_HIP_CHECK(hipGetLastError());
      g_kernel_launches += 1;
      BZ_HIP_CHECK(hipMemcpyAsync(partials.data(), d_partials,
                                  sizeof(fe) * static_cast<size_t>(blocks) * (kMaxDegree + 1),
                                  hipMemcpyDeviceToHost, ds->stream));
      BZ_HIP_CHECK(hipStreamSynchronize(ds->stream));
      for (u32 b = 0; b < blocks; ++b) {
        for (u32 k = 0; k < length; ++k) {
          poly[k] = fadd<F>(poly[k], partials[static_cast<size_t>(b) * (kMaxDegree + 1) + k]);
        }
      }
    } else {
      fe acc[kMaxDegree + 1];
      for (u32 k = 0; k <= kMaxDegree; ++k) acc[k] = F::zero();
      for (u64 i = 0; i < mid; ++i) {
        accumulate_row<F>(acc, h_mles.data(), n, mid, i, products.data(), d.num_products,
                          d.product_terms);
      }
      for (u32 k = 0; k < length; ++k) poly[k] = acc[k];
    }
    u8* out = polynomials + static_cast<size_t>(E::element_bytes) * length * round;
    for (u32 k = 0; k < length; ++k) E::store(out + E::element_bytes * k, poly[k]);
    // the caller's transcript draws the challenge (callback_sumcheck_transcript.h:27-45)
    u8* r_bytes = evaluation_point + static_cast<size_t>(E::element_bytes) * round;
    if (lease != nullptr) lease->unlock();
    reinterpret_cast<callback_t>(callback)(r_bytes, context, out, length);
    if (lease != nullptr) {
      lease->relock();
      if (on_device) ds->activate(); // the callback may have changed the thread's current device
    }
    if (round + 1 == num_variables) break;
    const fe r = E::load(r_bytes);
    const fe one_minus_r = fsub<F>(F::one(), r);
    if (on_device) {
      hipLaunchKernelGGL((k_sumcheck_fold<F>), dim3(ceil_div_u32(mid * d.num_mles, 256)), dim3(256),
                         0, ds->stream, d_next, d_mles, n, mid, d.num_mles, r, one_minus_r);
      BZ_HIP_CHECK(hipGetLastError());
      g_kernel_launches += 1;
      std::swap(d_mles, d_next);
    } else {
      h_next.assign(mid * d.num_mles, F::zero());
      for (u64 m = 0; m < d.num_mles; ++m) {
        for (u64 i = 0; i < mid; ++i) {
          fe v = F::mul(h_mles[m * n + i], one_minus_r);
          if (mid + i < n) v = fadd<F>(v, F::mul(r, h_mles[m * n + mid + i]));
          h_next[m * mid + i] = v;
        }
      }
      h_mles.swap(h_next);
    }
    n = mid;
  }
  if (on_device) {
    BZ_HIP_CHECK(hipStreamSynchronize(ds->stream));
    own.release();
  }
}
} // namespace

void prove_sumcheck(api_state& st, void* polynomials, void* evaluation_point, unsigned field_id,
                    const sumcheck_inputs& d, void* callback, void* context,
                    api_state::device_lease* lease) {
  BZ_RELEASE_ASSERT(d.n > 0, "sumcheck needs at least one row");
  BZ_RELEASE_ASSERT(d.round_degree >= 1 && d.round_degree <= kMaxDegree,
                    "round_degree must be in [1, 8]");
  BZ_RELEASE_ASSERT(d.n <= (1u << 30), "sumcheck tables are limited to 2^30 rows");
  if (field_id == 0) {
    prove<scalar25519_elements>(st, static_cast<u8*>(polynomials), static_cast<u8*>(evaluation_point),
                                d, callback, context, lease);
  } else if (field_id == 1) {
    prove<grumpkin_elements>(st, static_cast<u8*>(polynomials), static_cast<u8*>(evaluation_point), d,
                             callback, context, lease);
  } else {
    BZ_RELEASE_ASSERT(false, "unsupported field id");
  }
}
} // namespace bz::proof
