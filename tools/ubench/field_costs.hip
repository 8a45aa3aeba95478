// Cost of a field inversion in units of field products on gfx950, for the three Weierstrass base
// fields (field/mont29.h), every lane working on its own element with every SIMD busy.  This is
// the number that decides whether batch-affine bucket accumulation (affine + affine additions
// sharing one inversion through Montgomery's trick: 6 products per addition + inversion / batch)
// can beat the complete projective mixed addition (9.5 products): DESIGN.md section 11.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I. -o field_costs tools/ubench/field_costs.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#include "blitzar_amd/csrc/curve/sw29.h"

using namespace bz;

constexpr int kMulIters = 4096;
constexpr int kInvIters = 64;

template <class F> __device__ typename F::fe seed_element(u32 seed) {
  typename F::fe x = F::one();
  x.v[0] += seed * 2654435761u % 1000003u + threadIdx.x;
  x.v[1] += blockIdx.x;
  return F::reduce(F::norm(x));
}

template <class F> __global__ void __launch_bounds__(256) k_mul(u32* out, u32 seed) {
  typename F::fe x = seed_element<F>(seed), y = seed_element<F>(seed + 17);
  for (int i = 0; i < kMulIters; ++i) x = F::mul(x, y);
  out[blockIdx.x * 256 + threadIdx.x] = x.v[0] ^ x.v[F::N - 1];
}

template <class F> __global__ void __launch_bounds__(256) k_inv(u32* out, u32 seed) {
  typename F::fe x = seed_element<F>(seed);
  const typename F::fe y = seed_element<F>(seed + 17);
  for (int i = 0; i < kInvIters; ++i) x = F::add(F::invert(x), y);
  out[blockIdx.x * 256 + threadIdx.x] = x.v[0] ^ x.v[F::N - 1];
}

template <class G> __global__ void __launch_bounds__(256) k_add_mixed(u32* out, u32 seed) {
  using F = typename G::F;
  typename G::point p = G::identity();
  p.X = seed_element<F>(seed);
  typename G::affine q{seed_element<F>(seed + 3), seed_element<F>(seed + 5)};
  for (int i = 0; i < kMulIters / 8; ++i) p = G::add_mixed(p, q, ((i + threadIdx.x) & 1) != 0);
  out[blockIdx.x * 256 + threadIdx.x] = p.X.v[0] ^ p.Z.v[F::N - 1];
}

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) {                                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                                    \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

struct bench {
  const char* name;
  void (*fn)(u32*, u32);
  double ops;
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u32* d_out = nullptr;
  CHECK(hipMalloc(&d_out, sizeof(u32) * 256 * cus * 4));
  const bench benches[] = {
      {"bn254 mul", k_mul<bn254_fq29>, kMulIters},
      {"bn254 invert", k_inv<bn254_fq29>, kInvIters},
      {"bn254 add_mixed", k_add_mixed<bn254_g1_29>, kMulIters / 8},
      {"bls12-381 mul", k_mul<bls12_381_fp28>, kMulIters},
      {"bls12-381 invert", k_inv<bls12_381_fp28>, kInvIters},
      {"bls12-381 add_mixed", k_add_mixed<bls12_381_g1_28>, kMulIters / 8},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::printf("ns per operation per wave (every SIMD holding w waves), MI355X %d CUs\n", cus);
  std::printf("%-22s %12s %12s %12s\n", "operation", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD");
  double ns[6][3];
  int row = 0;
  for (const auto& b : benches) {
    std::printf("%-22s", b.name);
    int col = 0;
    for (int wps : {1, 2, 3}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 2u + r);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      // time one SIMD spends per wave-operation
      ns[row][col] = ms * 1e6 / 3.0 / (b.ops * wps);
      std::printf(" %12.1f", ns[row][col]);
      ++col;
    }
    std::printf("\n");
    ++row;
  }
  std::printf("inversion / product: bn254 %.1f  bls12-381 %.1f   mixed addition / product: bn254 "
              "%.2f  bls12-381 %.2f  (at 3 waves per SIMD)\n",
              ns[1][2] / ns[0][2], ns[4][2] / ns[3][2], ns[2][2] / ns[0][2], ns[5][2] / ns[3][2]);
  CHECK(hipFree(d_out));
  return 0;
}
