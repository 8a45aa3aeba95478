// What does rocprofv3's FETCH_SIZE count on gfx950 for the access pattern of k_accumulate -- a lane
// gathering one whole row of 64 / 96 / 128 bytes (bn254 + grumpkin / bls12-381 / curve25519 addends)
// at a random index with 16-byte loads -- against the wide coalesced streaming read the guide's
// "x 2" was calibrated on?  Known byte counts per launch are printed as JSON; run it under
//
//   rocprofv3 --pmc FETCH_SIZE -d <dir> -- tools/ubench/bin/gather_fetch
//
// and feed <dir> + this program's stdout to tools/prof/fetch_calibration.py, which writes the factor
// (known bytes / FETCH_SIZE bytes) per pattern into profiles/fetch_calibration.json.  Two table sizes:
// 64 MiB (inside the 256 MiB Infinity Cache after the first pass) and 2 GiB (beyond it).
// Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/gather_fetch tools/ubench/gather_fetch.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) {                                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                                    \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

constexpr int kGathersPerLane = 64;
constexpr int kBlocks = 1024, kThreads = 256;

// TableLog2 only makes the kernel NAME unique per table size (one PMC row per pattern)
template <int RowBytes, int TableLog2>
__global__ void __launch_bounds__(kThreads) k_gather(uint32_t* out, const uint8_t* table, uint64_t rows,
                                                     uint32_t seed) {
  uint64_t x = (static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x) * 0x9e3779b97f4a7c15ull + seed;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int g = 0; g < kGathersPerLane; ++g) {
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    const uint64_t row = ((x * 0x2545f4914f6cdd1dull) >> 20) % rows;
    const uint4* p = reinterpret_cast<const uint4*>(table + row * RowBytes);
#pragma unroll
    for (int i = 0; i < RowBytes / 16; ++i) {
      const uint4 v = p[i];
      acc.x ^= v.x;
      acc.y += v.y;
      acc.z ^= v.z;
      acc.w += v.w;
    }
  }
  out[blockIdx.x * kThreads + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// the guide's reference pattern: every lane 16 bytes, consecutive across the lanes
__global__ void __launch_bounds__(kThreads) k_stream(uint32_t* out, const uint4* src, uint64_t vectors) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; i < vectors;
       i += static_cast<uint64_t>(gridDim.x) * kThreads) {
    const uint4 v = src[i];
    acc.x ^= v.x;
    acc.y += v.y;
    acc.z ^= v.z;
    acc.w += v.w;
  }
  out[blockIdx.x * kThreads + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int RowBytes, int TableLog2> int run(uint32_t* d_out, const uint8_t* d_table, bool first) {
  const uint64_t rows = (uint64_t{1} << TableLog2) / RowBytes;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // one untimed pass (brings a table that fits into the Infinity Cache), then three measured ones
  hipLaunchKernelGGL((k_gather<RowBytes, TableLog2>), dim3(kBlocks), dim3(kThreads), 0, 0, d_out, d_table,
                     rows, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL((k_gather<RowBytes, TableLog2>), dim3(kBlocks), dim3(kThreads), 0, 0, d_out,
                       d_table, rows, 2u + r);
  }
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double gathers = static_cast<double>(kBlocks) * kThreads * kGathersPerLane;
  std::printf("%s{\"kernel\": \"k_gather<%d, %d>\", \"row_bytes\": %d, \"table_bytes\": %llu, "
              "\"known_bytes_per_launch\": %.0f, \"gathers_per_launch\": %.0f, \"launches\": 4, "
              "\"ms_per_launch\": %.4f, \"g_rows_per_s\": %.2f}",
              first ? "" : ",\n ", RowBytes, TableLog2, RowBytes,
              static_cast<unsigned long long>(uint64_t{1} << TableLog2), gathers * RowBytes, gathers,
              ms / 3, gathers / (ms / 3 * 1e-3) * 1e-9);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 0;
}

int main() {
  const uint64_t big = uint64_t{1} << 31;
  uint8_t* d_table = nullptr;
  uint32_t* d_out = nullptr;
  CHECK(hipMalloc(&d_table, big + 256));
  CHECK(hipMalloc(&d_out, sizeof(uint32_t) * kBlocks * 4 * kThreads)); // (k_stream: 4 x the blocks)
  // (in pieces: a single fill of 2 GiB and more faulted on this runtime)
  for (uint64_t off = 0; off < big; off += uint64_t{1} << 29) {
    CHECK(hipMemset(d_table + off, 0x5a, uint64_t{1} << 29));
  }
  CHECK(hipDeviceSynchronize());
  std::printf("[");
  if (run<128, 26>(d_out, d_table, true)) return 1;
  if (run<96, 26>(d_out, d_table, false)) return 1;
  if (run<64, 26>(d_out, d_table, false)) return 1;
  if (run<128, 31>(d_out, d_table, false)) return 1;
  if (run<96, 31>(d_out, d_table, false)) return 1;
  if (run<64, 31>(d_out, d_table, false)) return 1;
  // streaming reference: 1 GiB, twice
  const uint64_t vectors = (uint64_t{1} << 30) / 16;
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(k_stream, dim3(kBlocks * 4), dim3(kThreads), 0, 0, d_out,
                       reinterpret_cast<const uint4*>(d_table), vectors);
  }
  CHECK(hipDeviceSynchronize());
  std::printf(",\n {\"kernel\": \"k_stream\", \"known_bytes_per_launch\": %.0f, \"launches\": 2}]\n",
              static_cast<double>(vectors) * 16);
  CHECK(hipFree(d_table));
  CHECK(hipFree(d_out));
  return 0;
}
