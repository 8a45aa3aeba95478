// What a drop-in caller sees: the blocking sxt_* Pedersen entry points with HOST buffers
// (PCIe-inclusive), warm, through the public C ABI only -- the measurement behind bench.py's
// `host_api` block (the reference's own definition of the same quantity is its benchmark CLI,
// benchmark/multi_commitment/benchmark.m.cc:204-236, cloned in tools/multi_commitment; this driver adds
// warm-up calls, the built-in generators and a result check).
//
//   hostapi_bench [--log2n 20] [--samples 10] [--warmup 2] [--only COLUMNS caller|builtin]
//   (--only: that one case, for a rocprofv3 timeline of it: tools/prof/hostapi_timeline.py)
//
// For 1 and 10 columns of 2^log2n 32-byte scalars (std::mt19937{0} bytes, column-major, top nibble
// masked: BASELINE configs[1]'s scalars) it times
//   sxt_curve25519_compute_pedersen_commitments_with_generators   (generators uploaded per call)
//   sxt_curve25519_compute_pedersen_commitments                   (built-in generators, resident)
// and prints one JSON line; the two must agree on every commitment.
//
//   hipcc -O2 -std=c++17 -I include tools/pipeline_bench/hostapi_bench.cc -L blitzar_amd/lib \
//       -lblitzar_amd -Wl,-rpath,$PWD/blitzar_amd/lib -o tools/pipeline_bench/_build/hostapi_bench
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "blitzar_api.h"

static double now_ms() {
  using clock = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  unsigned log2n = 20, samples = 10, warmup = 2, only_columns = 0;
  int only_builtin = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&] { return static_cast<unsigned>(std::atoi(argv[++i])); };
    if (a == "--log2n") log2n = next();
    else if (a == "--samples") samples = next();
    else if (a == "--warmup") warmup = next();
    else if (a == "--only") {
      only_columns = next();
      only_builtin = std::string(argv[++i]) == "builtin" ? 1 : 0;
    }
    else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  const uint64_t n = uint64_t{1} << log2n;
  const unsigned max_columns = 10;
  const sxt_config config{SXT_GPU_BACKEND, n};
  if (sxt_init(&config) != 0) return 2;
  std::vector<sxt_ristretto255> generators(n);
  if (sxt_ristretto255_get_generators(generators.data(), n, 0) != 0) return 2;
  std::vector<uint8_t> data(static_cast<size_t>(max_columns) * n * 32);
  std::mt19937 gen{0};
  std::uniform_int_distribution<uint8_t> distribution(0, UINT8_MAX);
  for (auto& b : data) b = distribution(gen);
  for (size_t r = 0; r < static_cast<size_t>(max_columns) * n; ++r) data[r * 32 + 31] &= 0x0f;

  std::printf("{\"log2n\": %u, \"samples\": %u, \"warmup\": %u, \"cases\": [", log2n, samples, warmup);
  bool first = true, all_agree = true;
  for (unsigned columns : {1u, max_columns}) {
    if (only_columns != 0 && columns != only_columns) continue;
    std::vector<sxt_sequence_descriptor> desc(columns);
    for (unsigned c = 0; c < columns; ++c) {
      desc[c] = sxt_sequence_descriptor{32, n, data.data() + static_cast<size_t>(c) * n * 32, 0};
    }
    std::vector<sxt_ristretto255_compressed> with_caller(columns), with_builtin(columns);
    for (int builtin = 0; builtin < 2; ++builtin) {
      if (only_builtin >= 0 && builtin != only_builtin) continue;
      auto call = [&] {
        if (builtin) {
          sxt_curve25519_compute_pedersen_commitments(with_builtin.data(), columns, desc.data(), 0);
        } else {
          sxt_curve25519_compute_pedersen_commitments_with_generators(
              with_caller.data(), columns, desc.data(), generators.data());
        }
      };
      for (unsigned k = 0; k < warmup; ++k) call();
      std::vector<double> ms(samples);
      for (unsigned k = 0; k < samples; ++k) {
        const double t0 = now_ms();
        call();
        ms[k] = now_ms() - t0;
      }
      std::sort(ms.begin(), ms.end());
      double mean = 0;
      for (double v : ms) mean += v / samples;
      std::printf("%s{\"columns\": %u, \"generators\": \"%s\", \"ms_mean\": %.4f, \"ms_min\": %.4f, "
                  "\"ms_median\": %.4f, \"exponentiations_per_s\": %.4e, \"host_bytes_per_call\": %llu}",
                  first ? "" : ", ", columns, builtin ? "built-in (resident)" : "caller (uploaded per call)",
                  mean, ms.front(), ms[samples / 2], static_cast<double>(columns) * n / (mean * 1e-3),
                  static_cast<unsigned long long>(static_cast<uint64_t>(columns) * n * 32 +
                                                  (builtin ? 0 : n * sizeof(sxt_ristretto255))));
      first = false;
    }
    if (only_builtin < 0) {
      all_agree = all_agree && std::memcmp(with_caller.data(), with_builtin.data(), 32 * columns) == 0;
    }
  }
  std::printf("], \"caller_and_builtin_generators_agree\": %s}\n", all_agree ? "true" : "false");
  return all_agree ? 0 : 1;
}
