// Native driver of the throughput mode (include/blitzar_amd.h: bzamd_pipeline_next / _flush) for A/B
// runs on the GPU box: no Python, no torch import -- a variant costs a second or two of box time.
//
//   pipeline_bench [--curve 0..3] [--log2n 20] [--columns 1] [--steps 200] [--warmup 10]
//                  [--nbytes 32] [--null-stream] [--resident] [--skew] [--window-bits c]
//   --skew: two rows in three hold the same scalar (oversized bucket groups: the chunked sort path)
//
// One step = one bzamd_msm_device call of `columns` columns of 2^log2n uniform scalars (xorshift
// bytes; 32-byte columns masked to 252 bits) against caller generators: curve25519 the built-in
// generator sequence, the other curves g_i = (i + 1) G (bzamd_generator_multiples_device).  Prints one
// JSON line: ms per step in sequence, ms per lone call, the six stage times of both modes and a
// hash of the commitments (all steps must agree; variants of one configuration must print the
// same hash).
//
//   hipcc -O2 -std=c++17 -I include tools/pipeline_bench/pipeline_bench.cc -L blitzar_amd/lib \
//       -lblitzar_amd -Wl,-rpath,$PWD/blitzar_amd/lib -o tools/pipeline_bench/_build/pipeline_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "blitzar_amd.h"
#include "base_points.inc"

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

int main(int argc, char** argv) {
  unsigned curve = 0, log2n = 20, columns = 1, steps = 200, warmup = 10, nbytes = 32;
  bool null_stream = false, resident = false, skew = false, no_mask = false;
  unsigned window_bits = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&] { return static_cast<unsigned>(std::atoi(argv[++i])); };
    if (a == "--curve") curve = next();
    else if (a == "--log2n") log2n = next();
    else if (a == "--columns") columns = next();
    else if (a == "--steps") steps = next();
    else if (a == "--warmup") warmup = next();
    else if (a == "--nbytes") nbytes = next();
    else if (a == "--null-stream") null_stream = true;
    else if (a == "--resident") resident = true;
    else if (a == "--skew") skew = true;
    else if (a == "--no-mask") no_mask = true; // full 256-bit scalars: the top window is as full as the others
    else if (a == "--window-bits") window_bits = next(); // pin the window width (bzamd_set_window_bits)
    else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  const uint64_t n = uint64_t{1} << log2n;
  const sxt_config config{SXT_GPU_BACKEND, 0};
  if (sxt_init(&config) != 0) return 2;
  if (window_bits != 0) bzamd_set_window_bits(window_bits);
  hipStream_t stream = nullptr;
  if (!null_stream) CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  static const size_t gen_size[4] = {160, 104, 72, 72}, out_size[4] = {32, 48, 72, 72};
  // scalars: xorshift64* bytes, top nibble masked for 32-byte columns (252-bit scalars)
  std::vector<uint8_t> host(static_cast<size_t>(columns) * n * nbytes);
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (size_t i = 0; i + 8 <= host.size(); i += 8) {
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    const uint64_t v = x * 0x2545f4914f6cdd1dull;
    std::memcpy(&host[i], &v, 8);
  }
  if (nbytes == 32 && !no_mask) {
    for (size_t r = 0; r < static_cast<size_t>(columns) * n; ++r) host[r * 32 + 31] &= 0x0f;
  }
  if (skew) {
    for (size_t r = 0; r < static_cast<size_t>(columns) * n; ++r) {
      if (r % 3 != 0) std::memcpy(&host[r * nbytes], &host[0], nbytes);
    }
  }
  uint8_t* d_scalars = nullptr;
  CHECK(hipMalloc(&d_scalars, host.size()));
  CHECK(hipMemcpy(d_scalars, host.data(), host.size(), hipMemcpyHostToDevice));
  void* d_gens = nullptr;
  CHECK(hipMalloc(&d_gens, gen_size[curve] * n));
  if (curve == 0) {
    bzamd_ristretto255_generators_device(static_cast<sxt_ristretto255*>(d_gens), 0, n, stream);
  } else {
    // g_i = (i + 1) G, G = the reference's generate_random_element(rng{1, 2}) (base_points.inc)
    const unsigned char* base = curve == 1 ? kBasePoint1 : (curve == 2 ? kBasePoint2 : kBasePoint3);
    void* d_base = nullptr;
    CHECK(hipMalloc(&d_base, gen_size[curve]));
    CHECK(hipMemcpy(d_base, base, gen_size[curve], hipMemcpyHostToDevice));
    bzamd_generator_multiples_device(curve, d_gens, d_base, n, stream);
  }
  CHECK(hipStreamSynchronize(stream));

  std::vector<sxt_sequence_descriptor> desc(columns);
  for (unsigned c = 0; c < columns; ++c) {
    desc[c] = sxt_sequence_descriptor{static_cast<uint8_t>(nbytes), n,
                                      d_scalars + static_cast<size_t>(c) * n * nbytes, 0};
  }
  const size_t out_bytes = out_size[curve] * columns;
  const unsigned slots = steps > warmup ? steps : warmup;
  uint8_t* d_out = nullptr;
  CHECK(hipMalloc(&d_out, out_bytes * (slots + 1)));
  CHECK(hipMemset(d_out, 0, out_bytes * (slots + 1)));
  bzamd_generators* handle = nullptr;
  if (resident) handle = bzamd_generators_new_device(curve, d_gens, n, stream);

  auto call = [&](unsigned slot) {
    if (resident) {
      bzamd_msm_device_resident(d_out + out_bytes * slot, columns, desc.data(), handle, stream);
    } else {
      bzamd_msm_device(curve, d_out + out_bytes * slot, columns, desc.data(), d_gens, stream);
    }
  };
  // sequence (throughput mode)
  for (unsigned k = 0; k < warmup; ++k) {
    bzamd_pipeline_next();
    call(k);
  }
  bzamd_pipeline_flush(stream);
  CHECK(hipDeviceSynchronize());
  bzamd_stage_timing_begin_masked(steps, 1u << 3);
  const double t0 = now_ms();
  for (unsigned k = 0; k < steps; ++k) {
    bzamd_pipeline_next();
    call(k);
  }
  const double t_enqueued = now_ms();
  bzamd_pipeline_flush(stream);
  CHECK(hipDeviceSynchronize());
  const double seq_ms = (now_ms() - t0) / steps;
  const double host_ms = (t_enqueued - t0) / steps;
  double acc_in_seq[6];
  bzamd_stage_timing_collect(acc_in_seq);
  // all stage times in sequence (event pairs add bubbles: informational)
  bzamd_stage_timing_begin(steps);
  for (unsigned k = 0; k < steps; ++k) {
    bzamd_pipeline_next();
    call(k);
  }
  bzamd_pipeline_flush(stream);
  CHECK(hipDeviceSynchronize());
  double seq_stages[6];
  bzamd_stage_timing_collect(seq_stages);
  // lone calls
  const unsigned lone_steps = steps < 50 ? steps : 50;
  for (unsigned k = 0; k < 3; ++k) call(slots);
  CHECK(hipDeviceSynchronize());
  const double t1 = now_ms();
  for (unsigned k = 0; k < lone_steps; ++k) call(slots);
  CHECK(hipDeviceSynchronize());
  const double lone_ms = (now_ms() - t1) / lone_steps;
  bzamd_stage_timing_begin(lone_steps);
  for (unsigned k = 0; k < lone_steps; ++k) call(slots);
  CHECK(hipDeviceSynchronize());
  double lone_stages[6];
  bzamd_stage_timing_collect(lone_stages);

  std::vector<uint8_t> outs(out_bytes * (slots + 1));
  CHECK(hipMemcpy(outs.data(), d_out, outs.size(), hipMemcpyDeviceToHost));
  bool same = true;
  for (unsigned k = 1; k < steps; ++k) {
    same = same && std::memcmp(&outs[0], &outs[out_bytes * k], out_bytes) == 0;
  }
  same = same && std::memcmp(&outs[0], &outs[out_bytes * slots], out_bytes) == 0;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < out_bytes; ++i) h = (h ^ outs[i]) * 0x100000001b3ull;
  std::printf("{\"curve\": %u, \"log2n\": %u, \"columns\": %u, \"steps\": %u, \"ms_per_step\": %.4f, "
              "\"host_enqueue_ms\": %.4f, \"lone_ms\": %.4f, \"acc_in_sequence_ms\": %.4f, ",
              curve, log2n, columns, steps, seq_ms, host_ms, lone_ms, acc_in_seq[3] / steps);
  std::printf("\"sequence_stage_ms\": [");
  for (int s = 0; s < 6; ++s) std::printf("%s%.4f", s ? ", " : "", seq_stages[s] / steps);
  std::printf("], \"lone_stage_ms\": [");
  for (int s = 0; s < 6; ++s) std::printf("%s%.4f", s ? ", " : "", lone_stages[s] / lone_steps);
  std::printf("], \"outputs_agree\": %s, \"hash\": \"%016llx\"}\n", same ? "true" : "false",
              static_cast<unsigned long long>(h));
  if (handle != nullptr) bzamd_generators_free(handle);
  return same ? 0 : 1;
}
