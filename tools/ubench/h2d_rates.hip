// Host-to-device copy rates of the box, for the design of the blocking sxt_* entry points (api/capi.hip:
// every call uploads its scalars, and the caller's generators, from PAGEABLE host memory).  Not part of
// the product.
//
//   hipcc --offload-arch=gfx950 -O2 -pthread -o h2d_rates tools/ubench/h2d_rates.hip && ./h2d_rates
//
// Measures, for a 192 MiB buffer (BASELINE configs[1]: 32 MiB of scalars + 160 MiB of generators):
//   * hipMemcpyAsync from pageable memory (what round 3 does), one call and in 8 MiB pieces;
//   * hipMemcpyAsync from pinned memory, one stream and two streams;
//   * host memcpy pageable -> pinned with 1 .. 16 threads (the staging copy of a pinned ring);
//   * the two together: a ring of pinned slots filled by T host threads and drained by the DMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) {                                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                                    \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

static double now_ms() {
  using clock = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clock::now().time_since_epoch()).count();
}

static void parallel_copy(uint8_t* dst, const uint8_t* src, size_t bytes, unsigned threads) {
  if (threads <= 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> workers;
  const size_t piece = (bytes / threads + 4095) & ~size_t{4095};
  for (unsigned t = 0; t < threads; ++t) {
    const size_t lo = std::min(bytes, t * piece), hi = std::min(bytes, lo + piece);
    if (lo < hi) workers.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& w : workers) w.join();
}

int main() {
  const size_t bytes = size_t{192} << 20;
  std::vector<uint8_t> pageable(bytes);
  for (size_t i = 0; i < bytes; i += 4096) pageable[i] = static_cast<uint8_t>(i >> 12);
  uint8_t* pinned = nullptr;
  uint8_t* dev = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&pinned), bytes, hipHostMallocDefault));
  CHECK(hipMalloc(reinterpret_cast<void**>(&dev), bytes));
  std::memcpy(pinned, pageable.data(), bytes);
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  auto report = [&](const char* what, double ms) {
    std::printf("%-64s %7.3f ms  %6.1f GB/s\n", what, ms, bytes / ms / 1e6);
  };
  const int reps = 5;
  for (int warm = 0; warm < 2; ++warm) {
    CHECK(hipMemcpyAsync(dev, pageable.data(), bytes, hipMemcpyHostToDevice, s0));
    CHECK(hipStreamSynchronize(s0));
  }
  double best = 1e9;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now_ms();
    CHECK(hipMemcpyAsync(dev, pageable.data(), bytes, hipMemcpyHostToDevice, s0));
    CHECK(hipStreamSynchronize(s0));
    best = std::min(best, now_ms() - t0);
  }
  report("pageable, one hipMemcpyAsync", best);
  best = 1e9;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now_ms();
    for (size_t off = 0; off < bytes; off += size_t{8} << 20) {
      CHECK(hipMemcpyAsync(dev + off, pageable.data() + off, size_t{8} << 20, hipMemcpyHostToDevice, s0));
    }
    CHECK(hipStreamSynchronize(s0));
    best = std::min(best, now_ms() - t0);
  }
  report("pageable, 8 MiB pieces on one stream", best);
  best = 1e9;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now_ms();
    CHECK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s0));
    CHECK(hipStreamSynchronize(s0));
    best = std::min(best, now_ms() - t0);
  }
  report("pinned, one hipMemcpyAsync", best);
  best = 1e9;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now_ms();
    CHECK(hipMemcpyAsync(dev, pinned, bytes / 2, hipMemcpyHostToDevice, s0));
    CHECK(hipMemcpyAsync(dev + bytes / 2, pinned + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s1));
    CHECK(hipStreamSynchronize(s0));
    CHECK(hipStreamSynchronize(s1));
    best = std::min(best, now_ms() - t0);
  }
  report("pinned, halves on two streams", best);
  for (unsigned threads : {1u, 2u, 4u, 8u, 16u}) {
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
      const double t0 = now_ms();
      parallel_copy(pinned, pageable.data(), bytes, threads);
      best = std::min(best, now_ms() - t0);
    }
    char what[96];
    std::snprintf(what, sizeof(what), "host memcpy pageable -> pinned, %u thread(s)", threads);
    report(what, best);
  }
  // ring: slots of `slot` bytes, filled by T threads (fresh threads per slot: an upper bound on the
  // cost of a worker pool), each slot's DMA alternating between the two streams
  for (size_t slot_mib : {4u, 8u, 16u}) {
    for (unsigned threads : {4u, 8u}) {
      const size_t slot = slot_mib << 20;
      const int slots = 6;
      uint8_t* ring = nullptr;
      CHECK(hipHostMalloc(reinterpret_cast<void**>(&ring), slot * slots, hipHostMallocDefault));
      hipEvent_t freed[6];
      for (auto& ev : freed) CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      best = 1e9;
      for (int r = 0; r < reps; ++r) {
        bool armed[6] = {};
        const double t0 = now_ms();
        int k = 0;
        for (size_t off = 0; off < bytes; off += slot, ++k) {
          const int s = k % slots;
          if (armed[s]) CHECK(hipEventSynchronize(freed[s]));
          const size_t len = std::min(slot, bytes - off);
          parallel_copy(ring + s * slot, pageable.data() + off, len, threads);
          hipStream_t st = (k & 1) ? s1 : s0;
          CHECK(hipMemcpyAsync(dev + off, ring + s * slot, len, hipMemcpyHostToDevice, st));
          CHECK(hipEventRecord(freed[s], st));
          armed[s] = true;
        }
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipStreamSynchronize(s1));
        best = std::min(best, now_ms() - t0);
      }
      char what[96];
      std::snprintf(what, sizeof(what), "pinned ring: %zu MiB slots, %u copy threads, two streams",
                    slot_mib, threads);
      report(what, best);
      for (auto& ev : freed) (void)hipEventDestroy(ev);
      (void)hipHostFree(ring);
    }
  }
  std::printf("host threads available: %u\n", std::thread::hardware_concurrency());
  return 0;
}
