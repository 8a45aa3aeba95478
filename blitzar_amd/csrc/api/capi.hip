// extern "C" surface: the drop-in Blitzar entry points (include/blitzar_api.h) and the
// device-resident extensions (include/blitzar_amd.h).
//
// Reference counterparts: cbindings/backend.cc:42-134 (sxt_init + backend singleton),
// cbindings/pedersen.cc:44-251 (descriptor validation + the five Pedersen entry points),
// cbindings/get_generators.cc:32-59, cbindings/get_one_commit.cc:29-42,
// cbindings/fixed_pedersen.cc:29-106.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "blitzar_amd/csrc/api/state.h"
#include "blitzar_amd/csrc/fixed/dump.h"
#include "blitzar_amd/csrc/fixed/handle.h"
#include "include/blitzar_amd.h"

using namespace bz;

namespace bz {
std::atomic<u64> g_kernel_launches{0};

namespace {
api_state* g_state = nullptr;

api_state& state() {
  BZ_RELEASE_ASSERT(g_state != nullptr, "backend not initialised (call sxt_init first)");
  return *g_state;
}

void init_host_generators(api_state& st, u64 n) {
  st.host_generators.resize(n);
  st.host_one_commits.resize(n);
  if (n == 0) return;
  if (st.backend == SXT_GPU_BACKEND) {
    // derive on the device (reference K15), keep both the raw p3 copy (served by
    // sxt_ristretto255_get_generators) and the resident addends
    BZ_RELEASE_ASSERT(curve25519_vtable().addend_size == sizeof(ed29_cached_packed),
                      "built-in generator derivation and MSM engine disagree on the addend layout");
    ed_point* d_raw = nullptr;
    BZ_HIP_CHECK(hipMalloc(&d_raw, sizeof(ed_point) * n));
    BZ_HIP_CHECK(hipMalloc(&st.d_builtin_addends, curve25519_vtable().resident_addend_size * n));
    builtin_generators_enqueue(d_raw, 0, n, st.stream);
    g_kernel_launches += 1;
    curve25519_vtable().prepare_resident(st.d_builtin_addends, d_raw, n, st.stream);
    g_kernel_launches += 1;
    BZ_HIP_CHECK(hipMemcpyAsync(st.host_generators.data(), d_raw, sizeof(ed_point) * n,
                                hipMemcpyDeviceToHost, st.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(st.stream));
    BZ_HIP_CHECK(hipFree(d_raw));
  } else {
    for (u64 i = 0; i < n; ++i) st.host_generators[i] = ed::base_element(i);
  }
  // one_commit table: entry i = g_0 + ... + g_{i-1}, accumulated left to right from the identity.
  // This is a serial dependency chain whose raw limbs are observable, so it stays on the host in
  // both backends, exactly like the reference (cpu_one_commitments.cc:29-40).
  ed_point prev = ed::identity();
  for (u64 i = 0; i < n; ++i) {
    st.host_one_commits[i] = prev;
    prev = ed::add(prev, st.host_generators[i]);
  }
}

// generators [offset, offset + n) as raw p3 on the host
void host_builtin_generators(api_state& st, ed_point* out, u64 n, u64 offset) {
  u64 done = 0;
  if (offset < st.host_generators.size()) {
    done = std::min<u64>(n, st.host_generators.size() - offset);
    std::copy_n(st.host_generators.begin() + offset, done, out);
  }
  if (done == n) return;
  const u64 first = offset + done, rest = n - done;
  if (st.backend == SXT_GPU_BACKEND) {
    st.io.reset(sizeof(ed_point) * rest + 256, st.stream);
    ed_point* d = st.io.take<ed_point>(rest);
    builtin_generators_enqueue(d, first, rest, st.stream);
    g_kernel_launches += 1;
    BZ_HIP_CHECK(hipMemcpyAsync(out + done, d, sizeof(ed_point) * rest, hipMemcpyDeviceToHost,
                                st.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(st.stream));
  } else {
    for (u64 i = 0; i < rest; ++i) out[done + i] = ed::base_element(first + i);
  }
}

struct checked_columns {
  std::vector<host_column> cols;
  u64 longest = 0;
  size_t total_bytes = 0;
};

// reference: populate_exponent_sequence, cbindings/pedersen.cc:44-68 (+ the signed-width rule of
// sxt/multiexp/pippenger/exponent_aggregates_computation.cc:99-101)
checked_columns check_descriptors(const sxt_sequence_descriptor* descriptors, u32 num_sequences) {
  BZ_RELEASE_ASSERT(descriptors != nullptr, "descriptors is null");
  checked_columns r;
  r.cols.resize(num_sequences);
  for (u32 i = 0; i < num_sequences; ++i) {
    const sxt_sequence_descriptor& d = descriptors[i];
    BZ_RELEASE_ASSERT(d.n == 0 || d.data != nullptr, "descriptor has n > 0 but null data");
    BZ_RELEASE_ASSERT(d.element_nbytes != 0 && d.element_nbytes <= 32,
                      "element_nbytes must be in [1, 32]");
    BZ_RELEASE_ASSERT(!d.is_signed || d.element_nbytes <= 16,
                      "signed sequences need element_nbytes <= 16");
    BZ_RELEASE_ASSERT(d.n < (uint64_t{1} << 31), "sequences are limited to 2^31 - 1 rows");
    r.cols[i] = byte_column(d.data, d.n, d.element_nbytes, d.is_signed != 0);
    r.longest = std::max<u64>(r.longest, d.n);
    r.total_bytes += device_arena::padded(static_cast<size_t>(d.n) * d.element_nbytes + 32);
  }
  return r;
}

enum class generator_source { host_api, builtin };

// the Pedersen path of all five entry points
void compute_commitments(const curve_vtable& vt, void* commitments, u32 num_sequences,
                         const sxt_sequence_descriptor* descriptors, const void* generators,
                         generator_source source, u64 offset_generators,
                         bool projective_out = false) {
  if (num_sequences == 0) return; // reference: returns before touching anything
  BZ_RELEASE_ASSERT(commitments != nullptr, "commitments is null");
  api_state& st = state();
  checked_columns cc = check_descriptors(descriptors, num_sequences);
  const u32 out_stride = static_cast<u32>(projective_out ? vt.projective_size : vt.output_size);

  if (st.backend == SXT_CPU_BACKEND) {
    std::vector<ed_point> builtin;
    const void* gens = generators;
    if (source == generator_source::builtin) {
      builtin.resize(cc.longest);
      host_builtin_generators(st, builtin.data(), cc.longest, offset_generators);
      gens = builtin.data();
    }
    vt.msm_host(static_cast<u8*>(commitments), out_stride, projective_out, cc.cols, gens, false,
                cc.longest);
    return;
  }

  // GPU backend: stage host operands into the io arena, run, copy the encodings back.  The
  // columns are uploaded in chunks on a copy stream while the engine works on the previous chunk
  // (the DMA engines and the CUs are independent): the reference benchmark's 10 x 2^20 x 32-byte
  // job spends a third of its time in H2D copies otherwise.
  st.activate();
  const size_t gen_bytes = source == generator_source::host_api
                               ? device_arena::padded(vt.api_generator_size * cc.longest + 32) +
                                     device_arena::padded(vt.addend_size * (cc.longest + 1))
                               : device_arena::padded(vt.addend_size * (cc.longest + 1));
  const size_t out_bytes = device_arena::padded(static_cast<size_t>(out_stride) * num_sequences);
  st.io.reset(cc.total_bytes + gen_bytes + out_bytes + 1024, st.stream);
  if (st.copy_stream == nullptr) {
    BZ_HIP_CHECK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking));
  }
  std::vector<hipEvent_t> events;
  auto signal = [&](hipStream_t from, hipStream_t to) {
    hipEvent_t e;
    BZ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BZ_HIP_CHECK(hipEventRecord(e, from));
    BZ_HIP_CHECK(hipStreamWaitEvent(to, e, 0));
    events.push_back(e);
  };
  // the io arena may have been reallocated on st.stream: order the copy stream behind it
  signal(st.stream, st.copy_stream);

  const void* d_addends = nullptr;
  bool resident = false;
  if (source == generator_source::host_api) {
    u8* d_api = st.io.take<u8>(vt.api_generator_size * cc.longest + 32);
    void* prepared = st.io.take<u8>(vt.addend_size * (cc.longest + 1));
    if (cc.longest > 0) {
      BZ_HIP_CHECK(hipMemcpyAsync(d_api, generators, vt.api_generator_size * cc.longest,
                                  hipMemcpyHostToDevice, st.copy_stream));
      signal(st.copy_stream, st.stream);
      vt.prepare_addends(prepared, d_api, cc.longest, st.stream);
      g_kernel_launches += 1;
    }
    d_addends = prepared;
  } else if (offset_generators + cc.longest <= st.host_generators.size() &&
             st.d_builtin_addends != nullptr) {
    d_addends = static_cast<const char*>(st.d_builtin_addends) +
                vt.resident_addend_size * offset_generators;
    resident = true;
  } else {
    ed29_cached_packed* d = st.io.take<ed29_cached_packed>(cc.longest + 1);
    builtin_addends_enqueue(d, offset_generators, cc.longest, st.stream);
    g_kernel_launches += 1;
    d_addends = d;
  }
  u8* d_out = st.io.take<u8>(static_cast<size_t>(out_stride) * num_sequences);

  constexpr size_t kChunkBytes = size_t{48} << 20;
  for (size_t begin = 0; begin < cc.cols.size();) {
    size_t end = begin, bytes_in_chunk = 0;
    while (end < cc.cols.size() && (end == begin || bytes_in_chunk < kChunkBytes)) {
      host_column& col = cc.cols[end];
      if (col.n == 0) {
        col.data = nullptr;
      } else {
        const size_t bytes = static_cast<size_t>(col.n) * col.row_stride;
        u8* d = st.io.take<u8>(bytes + 32);
        BZ_HIP_CHECK(hipMemcpyAsync(d, col.data, bytes, hipMemcpyHostToDevice, st.copy_stream));
        col.data = d;
        bytes_in_chunk += bytes;
      }
      ++end;
    }
    signal(st.copy_stream, st.stream);
    const std::vector<host_column> chunk(cc.cols.begin() + begin, cc.cols.begin() + end);
    u8* out_k = d_out + begin * static_cast<size_t>(out_stride);
    if (resident) {
      vt.msm_resident(*st.ctx, out_k, out_stride, projective_out, chunk, d_addends, st.stream);
    } else {
      vt.msm(*st.ctx, out_k, out_stride, projective_out, chunk, d_addends, nullptr, st.stream);
    }
    begin = end;
  }
  BZ_HIP_CHECK(hipMemcpyAsync(commitments, d_out, static_cast<size_t>(out_stride) * num_sequences,
                              hipMemcpyDeviceToHost, st.stream));
  BZ_HIP_CHECK(hipStreamSynchronize(st.stream));
  for (auto& e : events) (void)hipEventDestroy(e);
}

int backend_from_environment(int backend) {
  const char* val = std::getenv("BLITZAR_BACKEND");
  if (val == nullptr) return backend;
  std::string s{val};
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  if (s == "cpu") return SXT_CPU_BACKEND;
  if (s == "gpu") return SXT_GPU_BACKEND;
  std::fprintf(stderr, "blitzar_amd: invalid BLITZAR_BACKEND value %s\n", val);
  std::abort();
}
} // namespace

api_state* current_state() { return g_state; }
} // namespace bz

extern "C" {
//--------------------------------------------------------------------------------------------------
// sxt_init
//--------------------------------------------------------------------------------------------------
int sxt_init(const struct sxt_config* config) {
  BZ_RELEASE_ASSERT(config != nullptr, "config input to `sxt_init` is null");
  BZ_RELEASE_ASSERT(g_state == nullptr, "trying to reinitialize the backend in `sxt_init`");
  const int backend = backend_from_environment(config->backend);
  if (backend != SXT_GPU_BACKEND && backend != SXT_CPU_BACKEND) return 1;
  auto st = std::make_unique<api_state>();
  st->backend = backend;
  if (backend == SXT_GPU_BACKEND) {
    // no silent fallback: a GPU backend without a GPU is a hard error, as in the reference
    // (cbindings/backend.cc:61-63 "no supported GPUs found")
    BZ_RELEASE_ASSERT(device_count() > 0, "no supported GPUs found");
    BZ_HIP_CHECK(hipGetDevice(&st->device));
    BZ_HIP_CHECK(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
    st->ctx = msm_context_new();
  }
  init_host_generators(*st, config->num_precomputed_generators);
  g_state = st.release();
  return 0;
}

//--------------------------------------------------------------------------------------------------
// Pedersen commitments
//--------------------------------------------------------------------------------------------------
void sxt_curve25519_compute_pedersen_commitments(struct sxt_ristretto255_compressed* commitments,
                                                 uint32_t num_sequences,
                                                 const struct sxt_sequence_descriptor* descriptors,
                                                 uint64_t offset_generators) {
  compute_commitments(curve25519_vtable(), commitments, num_sequences, descriptors, nullptr,
                      generator_source::builtin, offset_generators);
}

void sxt_curve25519_compute_pedersen_commitments_with_generators(
    struct sxt_ristretto255_compressed* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_ristretto255* generators) {
  // null generators select the built-in ones at offset 0 (cbindings/pedersen.cc:92-97)
  compute_commitments(curve25519_vtable(), commitments, num_sequences, descriptors, generators,
                      generators == nullptr ? generator_source::builtin
                                            : generator_source::host_api,
                      0);
}

void sxt_bls12_381_g1_compute_pedersen_commitments_with_generators(
    struct sxt_bls12_381_g1_compressed* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_bls12_381_g1* generators) {
  if (num_sequences == 0) return;
  BZ_RELEASE_ASSERT(generators != nullptr, "generators is null");
  compute_commitments(bls12_381_vtable(), commitments, num_sequences, descriptors, generators,
                      generator_source::host_api, 0);
}

void sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators(
    struct sxt_bn254_g1* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_bn254_g1* generators) {
  if (num_sequences == 0) return;
  BZ_RELEASE_ASSERT(generators != nullptr, "generators is null");
  compute_commitments(bn254_vtable(), commitments, num_sequences, descriptors, generators,
                      generator_source::host_api, 0);
}

void sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators(
    struct sxt_grumpkin* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_grumpkin* generators) {
  if (num_sequences == 0) return;
  BZ_RELEASE_ASSERT(generators != nullptr, "generators is null");
  compute_commitments(grumpkin_vtable(), commitments, num_sequences, descriptors, generators,
                      generator_source::host_api, 0);
}

//--------------------------------------------------------------------------------------------------
// built-in generators
//--------------------------------------------------------------------------------------------------
int sxt_ristretto255_get_generators(struct sxt_ristretto255* generators, uint64_t num_generators,
                                    uint64_t offset_generators) {
  api_state& st = state();
  if (num_generators == 0) return 0;
  if (generators == nullptr) return 1;
  if (st.backend == SXT_GPU_BACKEND) st.activate();
  host_builtin_generators(st, reinterpret_cast<ed_point*>(generators), num_generators,
                          offset_generators);
  return 0;
}

int sxt_curve25519_get_one_commit(struct sxt_ristretto255* one_commit, uint64_t n) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(one_commit != nullptr, "one_commit is null");
  ed_point r;
  if (n < st.host_one_commits.size()) {
    r = st.host_one_commits[n];
  } else {
    // continue the left-to-right chain from the last cached prefix
    // (sxt/seqcommit/generator/precomputed_one_commitments.cc:56-70)
    u64 start = 0;
    r = ed::identity();
    if (!st.host_one_commits.empty()) {
      start = st.host_one_commits.size() - 1;
      r = st.host_one_commits[start];
    }
    if (st.backend == SXT_GPU_BACKEND) st.activate();
    std::vector<ed_point> gens(n - start);
    host_builtin_generators(st, gens.data(), n - start, start);
    for (const auto& g : gens) r = ed::add(r, g);
  }
  std::memcpy(one_commit, &r, sizeof(r));
  return 0;
}

//--------------------------------------------------------------------------------------------------
// fixed-base handles
//--------------------------------------------------------------------------------------------------
namespace {
unsigned partition_window_width() {
  // reference: sxt/multiexp/pippenger2/window_width.cc:30-44
  const char* val = std::getenv("BLITZAR_PARTITION_WINDOW_WIDTH");
  if (val == nullptr) return 16;
  const long w = std::strtol(val, nullptr, 10);
  BZ_RELEASE_ASSERT(w > 0 && w <= 32, "invalid BLITZAR_PARTITION_WINDOW_WIDTH");
  return static_cast<unsigned>(w);
}

void handle_make_resident(multiexp_handle& h) {
  api_state& st = state();
  if (st.backend != SXT_GPU_BACKEND || h.n == 0) return;
  st.activate();
  h.device = st.device;
  void* d_proj = nullptr;
  const size_t bytes = h.vt->projective_size * h.n;
  BZ_HIP_CHECK(hipMalloc(&d_proj, bytes));
  BZ_HIP_CHECK(hipMalloc(&h.d_addends, h.vt->resident_addend_size * (h.n + 1)));
  BZ_HIP_CHECK(hipMemcpyAsync(d_proj, h.host_projective.data(), bytes, hipMemcpyHostToDevice,
                              st.stream));
  h.vt->prepare_resident_projective(h.d_addends, d_proj, h.n, st.stream);
  g_kernel_launches += 1;
  BZ_HIP_CHECK(hipStreamSynchronize(st.stream));
  BZ_HIP_CHECK(hipFree(d_proj));
}

// the three fixed-base entry points differ only in how a row is cut into per-output bit fields
// 
void fixed_multiexponentiation(void* res, const multiexp_handle& h, const unsigned* bit_table,
                               unsigned uniform_bits, const unsigned* lengths,
                               unsigned num_outputs, unsigned n, const u8* scalars,
                               bool device_operands, hipStream_t caller_stream) {
  if (num_outputs == 0) return;
  BZ_RELEASE_ASSERT(res != nullptr, "res is null");
  api_state& st = state();
  u64 total_bits = 0;
  unsigned prev_len = 0, max_len = 0;
  std::vector<host_column> cols(num_outputs);
  for (unsigned k = 0; k < num_outputs; ++k) {
    const unsigned width = bit_table != nullptr ? bit_table[k] : uniform_bits;
    BZ_RELEASE_ASSERT(width > 0, "output bit width must be positive");
    BZ_RELEASE_ASSERT(width <= 256, "outputs wider than 256 bits are not supported");
    const unsigned len = lengths != nullptr ? lengths[k] : n;
    BZ_RELEASE_ASSERT(len >= prev_len, "output lengths must be sorted in ascending order");
    prev_len = len;
    max_len = std::max(max_len, len);
    cols[k] = host_column{scalars, len, 0, static_cast<u32>(total_bits), width, false};
    total_bits += width;
  }
  BZ_RELEASE_ASSERT(total_bits < (u64{1} << 32), "row too wide");
  const u64 row_bytes = (total_bits + 7) / 8;
  BZ_RELEASE_ASSERT(max_len <= h.n, "more rows than generators in the handle");
  BZ_RELEASE_ASSERT(max_len == 0 || scalars != nullptr, "scalars is null");
  for (auto& c : cols) {
    c.row_stride = row_bytes;
    // fold whole bytes of the bit offset into the base pointer
    c.data = scalars + (c.bit_offset >> 3);
    c.bit_offset &= 7;
  }
  const u32 out_stride = static_cast<u32>(h.vt->projective_size);

  // BLITZAR_DUMP_DIR: record packed / vlen calls with host operands (the plain byte-aligned entry
  // point is not recorded by the reference either, gpu_backend.cc:257-272)
  std::unique_ptr<dump_recorder> recorder;
  if (bit_table != nullptr && !device_operands) {
    recorder = std::make_unique<dump_recorder>(lengths != nullptr ? "vlen-multiexponentiation"
                                                                   : "packed-multiexponentiation");
    if (recorder->recording()) {
      recorder->write_inputs(h, bit_table, lengths, num_outputs, max_len, scalars,
                             static_cast<size_t>(row_bytes) * max_len);
    }
  }
  auto record_result = [&] {
    if (recorder && recorder->recording()) {
      recorder->write("result.bin", res, static_cast<size_t>(out_stride) * num_outputs);
    }
  };

  if (st.backend == SXT_CPU_BACKEND) {
    BZ_RELEASE_ASSERT(!device_operands, "device entry points need the GPU backend");
    h.vt->msm_host(static_cast<u8*>(res), out_stride, true, cols, h.host_projective.data(), true,
                   max_len);
    record_result();
    return;
  }
  if (device_operands) {
    h.vt->msm_resident(*st.context_for_current_device(), static_cast<u8*>(res), out_stride, true,
                       cols, h.d_addends, caller_stream);
    return;
  }
  st.activate();
  const size_t scalar_bytes = static_cast<size_t>(row_bytes) * max_len;
  const size_t out_bytes = static_cast<size_t>(out_stride) * num_outputs;
  st.io.reset(device_arena::padded(scalar_bytes + 64) + device_arena::padded(out_bytes) + 512,
              st.stream);
  u8* d_scalars = st.io.take<u8>(scalar_bytes + 64);
  if (scalar_bytes > 0) {
    BZ_HIP_CHECK(hipMemcpyAsync(d_scalars, scalars, scalar_bytes, hipMemcpyHostToDevice,
                                st.stream));
  }
  for (auto& c : cols) c.data = d_scalars + (c.data - scalars);
  u8* d_out = st.io.take<u8>(out_bytes);
  h.vt->msm_resident(*st.ctx, d_out, out_stride, true, cols, h.d_addends, st.stream);
  BZ_HIP_CHECK(hipMemcpyAsync(res, d_out, out_bytes, hipMemcpyDeviceToHost, st.stream));
  BZ_HIP_CHECK(hipStreamSynchronize(st.stream));
  record_result();
}
} // namespace

struct sxt_multiexp_handle* sxt_multiexp_handle_new(unsigned curve_id, const void* generators,
                                                    unsigned n) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(n == 0 || generators != nullptr, "generators is null");
  (void)state();
  auto h = std::make_unique<multiexp_handle>();
  h->vt = vt;
  h->n = n;
  h->window_width = partition_window_width();
  h->host_projective.assign(static_cast<const u8*>(generators),
                            static_cast<const u8*>(generators) + vt->projective_size * n);
  handle_make_resident(*h);
  return reinterpret_cast<sxt_multiexp_handle*>(h.release());
}

struct sxt_multiexp_handle* sxt_multiexp_handle_new_from_file(unsigned curve_id,
                                                              const char* filename) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  (void)state();
  std::FILE* f = std::fopen(filename, "rb");
  BZ_RELEASE_ASSERT(f != nullptr, "failed to open partition table file");
  auto h = std::make_unique<multiexp_handle>();
  h->vt = vt;
  const bool ok = vt->read_partition_generators(f, h->window_width, h->host_projective, h->n);
  std::fclose(f);
  BZ_RELEASE_ASSERT(ok, "malformed partition table file");
  handle_make_resident(*h);
  return reinterpret_cast<sxt_multiexp_handle*>(h.release());
}

void sxt_multiexp_handle_write_to_file(const struct sxt_multiexp_handle* handle,
                                       const char* filename) {
  const auto* h = reinterpret_cast<const multiexp_handle*>(handle);
  BZ_RELEASE_ASSERT(h != nullptr, "handle is null");
  std::FILE* f = std::fopen(filename, "wb");
  BZ_RELEASE_ASSERT(f != nullptr, "failed to open partition table file for writing");
  h->vt->write_partition_table(f, h->window_width, h->host_projective.data(), h->n);
  std::fclose(f);
}

void sxt_multiexp_handle_free(struct sxt_multiexp_handle* handle) {
  auto* h = reinterpret_cast<multiexp_handle*>(handle);
  if (h == nullptr) return;
  if (h->d_addends != nullptr) (void)hipFree(h->d_addends);
  delete h;
}

void sxt_fixed_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                   unsigned element_num_bytes, unsigned num_outputs, unsigned n,
                                   const uint8_t* scalars) {
  const auto* h = reinterpret_cast<const multiexp_handle*>(handle);
  BZ_RELEASE_ASSERT(h != nullptr, "handle is null");
  BZ_RELEASE_ASSERT(element_num_bytes > 0, "element_num_bytes must be positive");
  fixed_multiexponentiation(res, *h, nullptr, 8 * element_num_bytes, nullptr, num_outputs, n,
                            scalars, false, nullptr);
}

void sxt_fixed_packed_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                          const unsigned* output_bit_table, unsigned num_outputs,
                                          unsigned n, const uint8_t* scalars) {
  const auto* h = reinterpret_cast<const multiexp_handle*>(handle);
  BZ_RELEASE_ASSERT(h != nullptr, "handle is null");
  BZ_RELEASE_ASSERT(num_outputs == 0 || output_bit_table != nullptr, "output_bit_table is null");
  fixed_multiexponentiation(res, *h, output_bit_table, 0, nullptr, num_outputs, n, scalars, false,
                            nullptr);
}

void sxt_fixed_vlen_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                        const unsigned* output_bit_table,
                                        const unsigned* output_lengths, unsigned num_outputs,
                                        const uint8_t* scalars) {
  const auto* h = reinterpret_cast<const multiexp_handle*>(handle);
  BZ_RELEASE_ASSERT(h != nullptr, "handle is null");
  BZ_RELEASE_ASSERT(num_outputs == 0 || (output_bit_table != nullptr && output_lengths != nullptr),
                    "output_bit_table / output_lengths is null");
  fixed_multiexponentiation(res, *h, output_bit_table, 0, output_lengths, num_outputs, 0, scalars,
                            false, nullptr);
}

void bzamd_fixed_packed_multiexponentiation_device(void* res,
                                                   const struct sxt_multiexp_handle* handle,
                                                   const unsigned* output_bit_table,
                                                   const unsigned* output_lengths,
                                                   unsigned num_outputs, unsigned n,
                                                   const uint8_t* scalars, void* stream) {
  const auto* h = reinterpret_cast<const multiexp_handle*>(handle);
  BZ_RELEASE_ASSERT(h != nullptr, "handle is null");
  BZ_RELEASE_ASSERT(num_outputs == 0 || output_bit_table != nullptr, "output_bit_table is null");
  fixed_multiexponentiation(res, *h, output_bit_table, 0, output_lengths, num_outputs, n, scalars,
                            true, static_cast<hipStream_t>(stream));
}

//--------------------------------------------------------------------------------------------------
// out-of-scope provers: exported for link compatibility, abort when called
//--------------------------------------------------------------------------------------------------
void sxt_curve25519_prove_inner_product(struct sxt_ristretto255_compressed*,
                                        struct sxt_ristretto255_compressed*,
                                        struct sxt_curve25519_scalar*, struct sxt_transcript*,
                                        uint64_t, uint64_t, const struct sxt_curve25519_scalar*,
                                        const struct sxt_curve25519_scalar*) {
  std::fprintf(stderr, "blitzar_amd: sxt_curve25519_prove_inner_product is outside the MSM path "
                       "and not implemented\n");
  std::abort();
}

int sxt_curve25519_verify_inner_product(struct sxt_transcript*, uint64_t, uint64_t,
                                        const struct sxt_curve25519_scalar*,
                                        const struct sxt_curve25519_scalar*,
                                        const struct sxt_ristretto255*,
                                        const struct sxt_ristretto255_compressed*,
                                        const struct sxt_ristretto255_compressed*,
                                        const struct sxt_curve25519_scalar*) {
  std::fprintf(stderr, "blitzar_amd: sxt_curve25519_verify_inner_product is outside the MSM path "
                       "and not implemented\n");
  std::abort();
}

void sxt_prove_sumcheck(void*, void*, unsigned, const struct sumcheck_descriptor*, void*, void*) {
  std::fprintf(stderr,
               "blitzar_amd: sxt_prove_sumcheck is outside the MSM path and not implemented\n");
  std::abort();
}

//--------------------------------------------------------------------------------------------------
// extensions (include/blitzar_amd.h)
//--------------------------------------------------------------------------------------------------
const char* bzamd_version(void) { return "blitzar_amd 0.1 (gfx950)"; }

int bzamd_device_count(void) { return device_count(); }

int bzamd_active_backend(void) { return g_state == nullptr ? 0 : g_state->backend; }

uint64_t bzamd_kernel_launch_count(void) { return g_kernel_launches.load(); }

void bzamd_stage_timing_begin(uint64_t max_calls) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "stage timing needs the GPU backend");
  msm_context_timing_begin(st.context_for_current_device(), max_calls);
}

uint64_t bzamd_stage_timing_collect(double* out_ms) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "stage timing needs the GPU backend");
  return msm_context_timing_collect(st.context_for_current_device(), out_ms);
}

void bzamd_set_tuning(uint32_t max_window_bits, uint64_t max_tasks_per_batch,
                      uint64_t max_workspace_bytes) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "tuning applies to the GPU backend");
  msm_context_set_tuning(st.context_for_current_device(), max_window_bits, max_tasks_per_batch,
                         max_workspace_bytes);
}

void bzamd_reset_for_testing(void) {
  if (g_state == nullptr) return;
  delete g_state;
  g_state = nullptr;
}

namespace {
void msm_device(unsigned curve_id, void* out, uint32_t num_sequences,
                const struct sxt_sequence_descriptor* descriptors, const void* generators,
                void* stream, bool projective_out) {
  if (num_sequences == 0) return;
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(out != nullptr, "output is null");
  BZ_RELEASE_ASSERT(generators != nullptr, "generators is null");
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  checked_columns cc = check_descriptors(descriptors, num_sequences);
  vt->msm(*st.context_for_current_device(), static_cast<u8*>(out),
          static_cast<u32>(projective_out ? vt->projective_size : vt->output_size), projective_out,
          cc.cols, nullptr, generators, static_cast<hipStream_t>(stream));
}
} // namespace

void bzamd_msm_device(unsigned curve_id, void* commitments, uint32_t num_sequences,
                      const struct sxt_sequence_descriptor* descriptors, const void* generators,
                      void* stream) {
  msm_device(curve_id, commitments, num_sequences, descriptors, generators, stream, false);
}

void bzamd_msm_device_projective(unsigned curve_id, void* res, uint32_t num_sequences,
                                 const struct sxt_sequence_descriptor* descriptors,
                                 const void* generators, void* stream) {
  msm_device(curve_id, res, num_sequences, descriptors, generators, stream, true);
}

void bzamd_msm_projective(unsigned curve_id, void* res, uint32_t num_sequences,
                          const struct sxt_sequence_descriptor* descriptors,
                          const void* generators) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(num_sequences == 0 || generators != nullptr, "generators is null");
  compute_commitments(*vt, res, num_sequences, descriptors, generators, generator_source::host_api,
                      0, true);
}

void bzamd_fold_encode(unsigned curve_id, void* commitments, const void* partials,
                       uint32_t num_partials, uint32_t num_outputs) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  if (num_outputs == 0) return;
  BZ_RELEASE_ASSERT(commitments != nullptr, "commitments is null");
  BZ_RELEASE_ASSERT(num_partials == 0 || partials != nullptr, "partials is null");
  vt->fold_encode_host(static_cast<u8*>(commitments), partials, num_partials, num_outputs);
}

void bzamd_fold_encode_device(unsigned curve_id, void* commitments, const void* partials,
                              uint32_t num_partials, uint32_t num_outputs, void* stream) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  if (num_outputs == 0) return;
  BZ_RELEASE_ASSERT(commitments != nullptr, "commitments is null");
  BZ_RELEASE_ASSERT(num_partials == 0 || partials != nullptr, "partials is null");
  vt->fold_encode_device(static_cast<u8*>(commitments), partials, num_partials, num_outputs,
                         static_cast<hipStream_t>(stream));
  g_kernel_launches += 1;
}

struct bzamd_generators* bzamd_generators_new_device(unsigned curve_id, const void* generators,
                                                     uint64_t n, void* stream) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(n == 0 || generators != nullptr, "generators is null");
  auto g = std::make_unique<resident_generators>();
  g->vt = vt;
  g->n = n;
  BZ_HIP_CHECK(hipMalloc(&g->d_addends, vt->resident_addend_size * (n + 1)));
  vt->prepare_resident(g->d_addends, generators, n, static_cast<hipStream_t>(stream));
  g_kernel_launches += 1;
  BZ_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return reinterpret_cast<bzamd_generators*>(g.release());
}

struct bzamd_generators* bzamd_generators_new_host(unsigned curve_id, const void* generators,
                                                   uint64_t n) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(n == 0 || generators != nullptr, "generators is null");
  void* d_api = nullptr;
  BZ_HIP_CHECK(hipMalloc(&d_api, vt->api_generator_size * (n + 1)));
  BZ_HIP_CHECK(hipMemcpy(d_api, generators, vt->api_generator_size * n, hipMemcpyHostToDevice));
  auto* r = bzamd_generators_new_device(curve_id, d_api, n, nullptr);
  BZ_HIP_CHECK(hipFree(d_api));
  return r;
}

void bzamd_generators_free(struct bzamd_generators* gens) {
  auto* g = reinterpret_cast<resident_generators*>(gens);
  if (g == nullptr) return;
  if (g->d_addends != nullptr) (void)hipFree(g->d_addends);
  delete g;
}

void bzamd_msm_device_resident(void* commitments, uint32_t num_sequences,
                               const struct sxt_sequence_descriptor* descriptors,
                               const struct bzamd_generators* gens, void* stream) {
  if (num_sequences == 0) return;
  const auto* g = reinterpret_cast<const resident_generators*>(gens);
  BZ_RELEASE_ASSERT(g != nullptr, "generators handle is null");
  BZ_RELEASE_ASSERT(commitments != nullptr, "commitments is null");
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  checked_columns cc = check_descriptors(descriptors, num_sequences);
  BZ_RELEASE_ASSERT(cc.longest <= g->n, "sequence longer than the resident generator set");
  g->vt->msm_resident(*st.context_for_current_device(), static_cast<u8*>(commitments),
                      static_cast<u32>(g->vt->output_size), false, cc.cols, g->d_addends,
                      static_cast<hipStream_t>(stream));
}

void bzamd_generator_multiples_device(unsigned curve_id, void* generators, const void* base,
                                      uint64_t n, void* stream) {
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(n == 0 || (generators != nullptr && base != nullptr), "null operand");
  vt->generator_multiples(generators, base, n, static_cast<hipStream_t>(stream));
  g_kernel_launches += 1;
}

void bzamd_ristretto255_generators_device(struct sxt_ristretto255* generators, uint64_t first,
                                          uint64_t n, void* stream) {
  BZ_RELEASE_ASSERT(n == 0 || generators != nullptr, "generators is null");
  builtin_generators_enqueue(reinterpret_cast<ed_point*>(generators), first, n,
                             static_cast<hipStream_t>(stream));
  g_kernel_launches += 1;
}
} // extern "C"
