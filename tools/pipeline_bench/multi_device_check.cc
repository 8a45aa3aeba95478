// Self-check and timing of the one-process multi-device paths of the C ABI on whatever devices are
// visible (include/blitzar_amd.h: bzamd_msm_multi_device -- columns sharded over the devices, RCCL
// all-gather of the commitments -- and the blocking sxt_* entry points, which shard columns or split
// the rows of a long column over the devices).  Every sharded result is compared with the SAME
// computation confined to device 0 (whose parity with the reference is what tests/ establish), so
// the tool needs no oracle and runs anywhere: bench.py's rank 0 runs it in a subprocess on a
// multi-GPU node, `-m gpu` tests run it with BLITZAR_AMD_FORCE_SHARDS on one GPU.
//
//   multi_device_check [--log2n 18] [--columns 16] [--steps 5]
//
// Prints one JSON line; exit code 0 iff every comparison held.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "blitzar_amd.h"

#define CHECK(expr)                                                                      \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      std::fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e__));            \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

static double now_ms() {
  using clock = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clock::now().time_since_epoch()).count();
}

// column c: xorshift64* bytes seeded by c, top nibble masked (252-bit scalars)
static void fill_column(uint8_t* dst, uint64_t n, unsigned c) {
  uint64_t x = 0x9e3779b97f4a7c15ull ^ (0x632be59bd9b4e019ull * (c + 1));
  for (uint64_t i = 0; i < n * 4; ++i) {
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    const uint64_t v = x * 0x2545f4914f6cdd1dull;
    std::memcpy(dst + 8 * i, &v, 8);
  }
  for (uint64_t r = 0; r < n; ++r) dst[32 * r + 31] &= 0x0f;
}

int main(int argc, char** argv) {
  unsigned log2n = 18, columns = 16, steps = 5;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&] { return static_cast<unsigned>(std::atoi(argv[++i])); };
    if (a == "--log2n") log2n = next();
    else if (a == "--columns") columns = next();
    else if (a == "--steps") steps = next();
    else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  const uint64_t n = uint64_t{1} << log2n;
  const sxt_config config{SXT_GPU_BACKEND, 0};
  if (sxt_init(&config) != 0) return 2;
  const int D = bzamd_num_devices();
  const uint32_t per = bzamd_multi_device_columns_per_device(columns);
  std::vector<int> ids(D);
  for (int d = 0; d < D; ++d) ids[d] = bzamd_device_id(d);

  // host copies of every column; device copies on the owner (and all of them on slot 0)
  std::vector<std::vector<uint8_t>> host(columns, std::vector<uint8_t>(n * 32));
  for (unsigned c = 0; c < columns; ++c) fill_column(host[c].data(), n, c);
  std::vector<void*> gens(D, nullptr), out(D, nullptr);
  std::vector<uint8_t*> owned(columns, nullptr), on0(columns, nullptr);
  for (int d = 0; d < D; ++d) {
    CHECK(hipSetDevice(ids[d]));
    CHECK(hipMalloc(&gens[d], 160 * n));
    bzamd_ristretto255_generators_device(static_cast<sxt_ristretto255*>(gens[d]), 0, n, nullptr);
    CHECK(hipMalloc(&out[d], 32 * columns));
    CHECK(hipMemset(out[d], 0, 32 * columns));
  }
  for (unsigned c = 0; c < columns; ++c) {
    CHECK(hipSetDevice(ids[c / per]));
    CHECK(hipMalloc(&owned[c], n * 32));
    CHECK(hipMemcpy(owned[c], host[c].data(), n * 32, hipMemcpyHostToDevice));
    CHECK(hipSetDevice(ids[0]));
    CHECK(hipMalloc(&on0[c], n * 32));
    CHECK(hipMemcpy(on0[c], host[c].data(), n * 32, hipMemcpyHostToDevice));
  }
  for (int d = 0; d < D; ++d) {
    CHECK(hipSetDevice(ids[d]));
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipSetDevice(ids[0]));

  // the same columns confined to device 0
  std::vector<sxt_sequence_descriptor> desc0(columns), desc(columns), hdesc(columns);
  for (unsigned c = 0; c < columns; ++c) {
    desc0[c] = sxt_sequence_descriptor{32, n, on0[c], 0};
    desc[c] = sxt_sequence_descriptor{32, n, owned[c], 0};
    hdesc[c] = sxt_sequence_descriptor{32, n, host[c].data(), 0};
  }
  void* d_want = nullptr;
  CHECK(hipMalloc(&d_want, 32 * columns));
  bzamd_msm_device(0, d_want, columns, desc0.data(), gens[0], nullptr);
  CHECK(hipDeviceSynchronize());
  double t0 = now_ms();
  for (unsigned k = 0; k < steps; ++k) bzamd_msm_device(0, d_want, columns, desc0.data(), gens[0], nullptr);
  CHECK(hipDeviceSynchronize());
  const double single_ms = (now_ms() - t0) / steps;
  std::vector<uint8_t> want(32 * columns), got(32 * columns);
  CHECK(hipMemcpy(want.data(), d_want, want.size(), hipMemcpyDeviceToHost));

  // 1. device-resident columns sharded over the devices, all-gather of the commitments
  bzamd_msm_multi_device(0, out.data(), columns, desc.data(), gens.data());
  t0 = now_ms();
  for (unsigned k = 0; k < steps; ++k) bzamd_msm_multi_device(0, out.data(), columns, desc.data(), gens.data());
  const double multi_ms = (now_ms() - t0) / steps;
  bool multi_ok = true;
  for (int d = 0; d < D; ++d) {
    CHECK(hipSetDevice(ids[d]));
    CHECK(hipMemcpy(got.data(), out[d], got.size(), hipMemcpyDeviceToHost));
    multi_ok = multi_ok && got == want;
  }
  CHECK(hipSetDevice(ids[0]));

  // 2. the blocking drop-in entry point with host buffers: columns sharded over the devices
  std::vector<sxt_ristretto255> host_gens(n);
  sxt_ristretto255_get_generators(host_gens.data(), n, 0);
  bzamd_set_shard_min_bytes(0);
  std::fill(got.begin(), got.end(), 0);
  sxt_curve25519_compute_pedersen_commitments_with_generators(
      reinterpret_cast<sxt_ristretto255_compressed*>(got.data()), columns, hdesc.data(), host_gens.data());
  t0 = now_ms();
  sxt_curve25519_compute_pedersen_commitments_with_generators(
      reinterpret_cast<sxt_ristretto255_compressed*>(got.data()), columns, hdesc.data(), host_gens.data());
  const double host_sharded_ms = now_ms() - t0;
  const bool column_shard_ok = got == want;
  bzamd_set_shard_min_bytes(~uint64_t{0});
  t0 = now_ms();
  sxt_curve25519_compute_pedersen_commitments_with_generators(
      reinterpret_cast<sxt_ristretto255_compressed*>(got.data()), columns, hdesc.data(), host_gens.data());
  const double host_single_ms = now_ms() - t0;
  const bool host_single_ok = got == want;

  // 3. ... and one long column: its rows split over the devices, partials folded on device 0
  bzamd_set_shard_min_bytes(0);
  uint8_t row_got[32] = {0};
  sxt_curve25519_compute_pedersen_commitments_with_generators(
      reinterpret_cast<sxt_ristretto255_compressed*>(row_got), 1, hdesc.data(), host_gens.data());
  const bool row_split_ok = std::memcmp(row_got, want.data(), 32) == 0;

  const bool ok = multi_ok && column_shard_ok && host_single_ok && row_split_ok;
  std::printf("{\"devices\": %d, \"device_ids\": [", D);
  for (int d = 0; d < D; ++d) std::printf("%s%d", d ? ", " : "", ids[d]);
  std::printf("], \"exchange\": \"%s\", \"curve\": \"curve25519\", \"columns\": %u, \"rows\": %llu, "
              "\"columns_per_device\": %u, \"ms_per_call_device_resident_sharded\": %.3f, "
              "\"ms_per_call_device_0_only\": %.3f, \"ms_host_api_sharded\": %.3f, "
              "\"ms_host_api_device_0_only\": %.3f, \"device_resident_sharded_ok\": %s, "
              "\"host_api_column_shard_ok\": %s, \"host_api_row_split_ok\": %s, \"ok\": %s}\n",
              bzamd_multi_device_exchange(), columns, static_cast<unsigned long long>(n), per,
              multi_ms, single_ms, host_sharded_ms, host_single_ms, multi_ok ? "true" : "false",
              column_shard_ok ? "true" : "false", row_split_ok ? "true" : "false",
              ok ? "true" : "false");
  return ok ? 0 : 1;
}
