// Field / group-operation throughput on gfx950: cycles per operation per wave with every SIMD
// holding `w` resident waves.  Calibrates DESIGN.md's ALU model.  Build twice to compare the two
// code shapes of f29::mul (-DBZ_F29_MAD_MODE=0: hipcc's reassociated columns, =1: carry rides in
// the mad addend).
#include <hip/hip_runtime.h>

#include <cstdio>

#include "blitzar_amd/csrc/curve/ed29.h"

using namespace bz;

constexpr int kIters = 2000;

__global__ void __launch_bounds__(256) k_f51_mul(u64* out, u64 seed) {
  fe51 x = {{seed + threadIdx.x, 2, 3, 4, 5}}, y = {{7, seed, 9, 10, threadIdx.x}};
  for (int i = 0; i < kIters; ++i) x = f51::mul(x, y);
  out[blockIdx.x * 256 + threadIdx.x] = x.v[0] ^ x.v[4];
}
__global__ void __launch_bounds__(256) k_f29_mul(u64* out, u64 seed) {
  fe29 x = {{(u32)seed + threadIdx.x, 2, 3, 4, 5, 6, 7, 8, 9}};
  fe29 y = {{7, (u32)seed, 9, 10, threadIdx.x, 1, 2, 3, 4}};
  for (int i = 0; i < kIters; ++i) x = f29::mul(x, y);
  out[blockIdx.x * 256 + threadIdx.x] = x.v[0] ^ x.v[8];
}
__global__ void __launch_bounds__(256) k_f29_sq(u64* out, u64 seed) {
  fe29 x = {{(u32)seed + threadIdx.x, 2, 3, 4, 5, 6, 7, 8, 9}};
  for (int i = 0; i < kIters; ++i) x = f29::sq(x);
  out[blockIdx.x * 256 + threadIdx.x] = x.v[0] ^ x.v[8];
}
__global__ void __launch_bounds__(256) k_ed_add(u64* out, u64 seed) {
  ed_point p = ed::identity();
  p.X.v[0] = seed + threadIdx.x;
  ed_cached q = ed::to_cached(p);
  q.T2d.v[1] = threadIdx.x;
  for (int i = 0; i < kIters / 4; ++i) p = ed::to_point(ed::add_cached(p, q));
  out[blockIdx.x * 256 + threadIdx.x] = p.X.v[0] ^ p.T.v[4];
}
__global__ void __launch_bounds__(256) k_ed29_add(u64* out, u64 seed) {
  ed29_point p = ed29::identity();
  p.X.v[0] = (u32)seed + threadIdx.x;
  ed29_cached q = ed29::to_cached(p);
  q.T2d.v[1] = threadIdx.x;
  for (int i = 0; i < kIters / 4; ++i) p = ed29::add_cached(p, q, ((i + threadIdx.x) & 1) != 0);
  out[blockIdx.x * 256 + threadIdx.x] = p.X.v[0] ^ p.T.v[8];
}
__global__ void __launch_bounds__(256) k_ed29_dbl(u64* out, u64 seed) {
  ed29_point p = ed29::identity();
  p.X.v[0] = (u32)seed + threadIdx.x;
  for (int i = 0; i < kIters / 4; ++i) p = ed29::dbl(p);
  out[blockIdx.x * 256 + threadIdx.x] = p.X.v[0] ^ p.T.v[8];
}

struct bench {
  const char* name;
  void (*fn)(u64*, u64);
  int ops;
};

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  u64* d_out = nullptr;
  if (hipMalloc(&d_out, sizeof(u64) * 256 * cus * 8) != hipSuccess) return 1;
  const bench benches[] = {{"f51::mul", k_f51_mul, kIters},       {"f29::mul", k_f29_mul, kIters},
                           {"f29::sq", k_f29_sq, kIters},         {"ed::add (f51)", k_ed_add, kIters / 4},
                           {"ed29::add_cached", k_ed29_add, kIters / 4},
                           {"ed29::dbl", k_ed29_dbl, kIters / 4}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::printf("BZ_F29_MAD_MODE=%d  (nominal-clock cycles per op per wave; lower = faster)\n",
              BZ_F29_MAD_MODE);
  std::printf("%-20s %10s %10s %10s %10s   ns/op/lane-wave @4w\n", "op", "1 w/SIMD", "2 w/SIMD",
              "4 w/SIMD", "8 w/SIMD");
  for (const auto& b : benches) {
    std::printf("%-20s", b.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 2);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::printf(" %10.1f", ms * 1e-3 * clk / (static_cast<double>(b.ops) * wps));
    }
    std::printf("\n");
  }
  return 0;
}
