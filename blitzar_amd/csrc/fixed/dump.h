// BLITZAR_DUMP_DIR recording of fixed-base multiexponentiations, in the reference's on-disk layout
// (sxt/base/system/directory_recorder.cc:29-65: one numbered directory per call;
// sxt/multiexp/pippenger2/multiexponentiation_serialization.h:56-103: the files;
// sxt/cbindings/backend/gpu_backend.cc:286-301,317-332: what is recorded and when), so that calls
// captured from a CUDA deployment replay here and vice versa (tools/replay_dump.py):
//   <dir>/{packed,vlen}-multiexponentiation-<k>/
//     output_bit_table.bin  u32[num_outputs]
//     output_lengths.bin    u32[num_outputs]          (vlen only)
//     scalars.bin           n rows of ceil(sum bits / 8) bytes
//     meta.txt              element type / accessor type (typeid names) / num_outputs
//     generators.bin        compact_element[n]
//     window_width.bin      u64
//     result.bin            projective element[num_outputs]   (written after the computation)
#pragma once

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/stat.h>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/fixed/handle.h"

namespace bz {

class dump_recorder {
public:
  explicit dump_recorder(const char* base_name) {
    const char* dir = std::getenv("BLITZAR_DUMP_DIR");
    if (dir == nullptr || dir[0] == 0) return;
    static std::atomic<unsigned> counter{0};
    name_ = std::string(dir) + "/" + base_name + "-" + std::to_string(counter++);
    BZ_RELEASE_ASSERT(::mkdir(name_.c_str(), 0777) == 0, "failed to create the dump directory");
  }
  bool recording() const { return !name_.empty(); }

  void write(const char* file, const void* data, size_t bytes) const {
    const std::string path = name_ + "/" + file;
    std::FILE* f = std::fopen(path.c_str(), "wb");
    BZ_RELEASE_ASSERT(f != nullptr, "failed to open a dump file");
    if (bytes > 0) BZ_RELEASE_ASSERT(std::fwrite(data, 1, bytes, f) == bytes, "short dump write");
    std::fclose(f);
  }

  void write_inputs(const multiexp_handle& h, const unsigned* bit_table, const unsigned* lengths,
                    unsigned num_outputs, u64 n, const u8* scalars, size_t scalar_bytes) const {
    write("output_bit_table.bin", bit_table, sizeof(unsigned) * num_outputs);
    if (lengths != nullptr) write("output_lengths.bin", lengths, sizeof(unsigned) * num_outputs);
    write("scalars.bin", scalars, scalar_bytes);
    const std::string meta = std::string("element type: ") + h.vt->element_type_name +
                             "\naccessor type: " + h.vt->accessor_type_name +
                             "\nnum_outputs: " + std::to_string(num_outputs) + "\n";
    write("meta.txt", meta.data(), meta.size());
    {
      const std::string path = name_ + "/generators.bin";
      std::FILE* f = std::fopen(path.c_str(), "wb");
      BZ_RELEASE_ASSERT(f != nullptr, "failed to open a dump file");
      h.vt->write_compact_generators(f, h.host_projective.data(), n);
      std::fclose(f);
    }
    const uint64_t w = h.window_width;
    write("window_width.bin", &w, sizeof(w));
  }

private:
  std::string name_;
};
} // namespace bz
