// Command-line clone of the reference's benchmark/multi_commitment:
// same arguments, same input recipe, same output lines, so that existing scripts
// (benchmark/scripts/run_benchmarks.py) can drive this library unchanged.
//
//   benchmark <cpu|gpu> <n> <num_samples> <num_commitments> <element_nbytes> <verbose>
//
// Reference behaviour restated (benchmark/multi_commitment/benchmark.m.cc):
//   :58-89    arguments; element_nbytes == 0 selects boolean data stored in one byte
//   :136-165  data = std::mt19937{0} through uniform_int_distribution<uint8_t>, column c occupies
//             bytes [c n nbytes, (c+1) n nbytes); generators = compute_base_element(i)
//   :178-243  report: mean / standard deviation over the samples, n * commitments / mean
// Everything goes through the public C ABI (include/blitzar_api.h); the generators are passed by
// the caller on every call, exactly like the reference benchmark does.
//
//   g++ -O2 -std=c++17 -I include tools/multi_commitment/benchmark.cc \
//       -L blitzar_amd/lib -lblitzar_amd -Wl,-rpath,$PWD/blitzar_amd/lib -o multi_commitment
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "blitzar_api.h"

int main(int argc, char* argv[]) {
  if (argc < 7) {
    std::cerr << "Usage: benchmark <cpu|gpu> <n> <num_samples> <num_commitments> <element_nbytes> "
                 "<verbose>\n";
    return -1;
  }
  const std::string backend_str = argv[1];
  if (backend_str != "cpu" && backend_str != "gpu") {
    std::cerr << "invalid backend: " << backend_str << "\n";
    return -1;
  }
  const long commitment_length = std::atol(argv[2]);
  const int num_samples = std::atoi(argv[3]);
  const long num_commitments = std::atol(argv[4]);
  int element_nbytes = std::atoi(argv[5]);
  const bool verbose = std::string(argv[6]) == "1";
  if (num_commitments <= 0 || commitment_length <= 0 || element_nbytes > 32 || num_samples <= 0) {
    std::cerr << "Restriction: 1 <= num_commitments, 1 <= commitment_length, "
                 "1 <= element_nbytes <= 32\n";
    return -1;
  }
  bool is_boolean = false;
  if (element_nbytes == 0) {
    is_boolean = true;
    element_nbytes = 1;
  }
  const sxt_config config{backend_str == "gpu" ? SXT_GPU_BACKEND : SXT_CPU_BACKEND, 0};
  if (sxt_init(&config) != 0) return -1;

  const double table_size = (num_commitments * commitment_length * element_nbytes) / 1024.;
  std::cout << "===== benchmark results" << std::endl;
  std::cout << "backend : " << backend_str << std::endl;
  std::cout << "commitment length : " << commitment_length << std::endl;
  std::cout << "number of commitments : " << num_commitments << std::endl;
  std::cout << "element_nbytes : " << element_nbytes << std::endl;
  std::cout << "is boolean : " << is_boolean << std::endl;
  std::cout << "table_size (MB) : " << table_size << std::endl;
  std::cout << "num_exponentations : " << (num_commitments * commitment_length) << std::endl;
  std::cout << "********************************************" << std::endl;

  std::vector<sxt_ristretto255> generators(commitment_length);
  if (sxt_ristretto255_get_generators(generators.data(), commitment_length, 0) != 0) return -1;
  std::vector<uint8_t> data_table(static_cast<size_t>(commitment_length) * num_commitments *
                                  element_nbytes);
  std::mt19937 gen{0};
  std::uniform_int_distribution<uint8_t> distribution(0, is_boolean ? 1 : UINT8_MAX);
  for (auto& b : data_table) b = distribution(gen);
  std::vector<sxt_sequence_descriptor> descriptors(num_commitments);
  for (long c = 0; c < num_commitments; ++c) {
    descriptors[c].element_nbytes = static_cast<uint8_t>(element_nbytes);
    descriptors[c].n = commitment_length;
    descriptors[c].data = data_table.data() + c * commitment_length * element_nbytes;
    descriptors[c].is_signed = 0;
  }
  std::vector<sxt_ristretto255_compressed> commitments(num_commitments);

  std::vector<double> durations;
  double mean = 0;
  for (int i = 0; i < num_samples; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    sxt_curve25519_compute_pedersen_commitments_with_generators(
        commitments.data(), static_cast<uint32_t>(num_commitments), descriptors.data(),
        generators.data());
    const auto t1 = std::chrono::steady_clock::now();
    const double s = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() / 1e6;
    durations.push_back(s);
    mean += s / num_samples;
  }
  double deviation = 0;
  for (double d : durations) deviation += std::pow(d - mean, 2.);
  deviation = std::sqrt(deviation / num_samples);
  const double throughput = commitment_length * num_commitments / mean;
  std::cout << "compute duration (s) : " << std::fixed << mean << std::endl;
  std::cout << "compute std deviation (s) : " << std::fixed << deviation << std::endl;
  std::cout << "throughput (exponentiations / s) : " << std::scientific << throughput << std::endl;
  // extension (off unless asked for, so that the reference's output format stays what scripts parse):
  // the mean without the first, cold sample (device workspace allocation, clocks coming up)
  if (std::getenv("BLITZAR_AMD_CLI_WARM") != nullptr && num_samples > 1) {
    double warm = 0;
    for (int i = 1; i < num_samples; ++i) warm += durations[i] / (num_samples - 1);
    std::cout << "warm compute duration (s) : " << std::fixed << warm << std::endl;
  }
  if (verbose) {
    std::cout << "===== result\n";
    for (long c = 0; c < num_commitments; ++c) {
      std::cout << "commitment " << c << " = 0x";
      for (int b = 0; b < 32; ++b) {
        std::cout << std::hex << std::setw(2) << std::setfill('0')
                  << static_cast<int>(commitments[c].ristretto_bytes[b]);
      }
      std::cout << std::dec << std::endl;
    }
  }
  std::cout << "********************************************" << std::endl;
  return 0;
}
