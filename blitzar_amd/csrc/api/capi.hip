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
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "blitzar_amd/csrc/base/log.h"
#include "blitzar_amd/csrc/api/rccl_loader.h"
#include "blitzar_amd/csrc/api/state.h"
#include "blitzar_amd/csrc/fixed/dump.h"
#include "blitzar_amd/csrc/fixed/handle.h"
#include "blitzar_amd/csrc/proof/inner_product.h"
#include "blitzar_amd/csrc/proof/sumcheck.h"
#include "blitzar_amd/csrc/proof/transcript.h"
#include "include/blitzar_amd.h"

using namespace bz;

namespace bz {
std::atomic<u64> g_kernel_launches{0};

namespace {
// bzamd_pipeline_next: consumed by the next device entry point THIS THREAD calls -- never by an
// engine call a blocking sxt_* function makes internally (those read their results right away)
thread_local bool t_pipeline_next = false;
void apply_pipeline_request(msm_context* ctx) {
  if (!t_pipeline_next) return;
  t_pipeline_next = false;
  msm_context_defer_next_tail(ctx);
}
api_state* g_state = nullptr;

api_state& state() {
  BZ_RELEASE_ASSERT(g_state != nullptr, "backend not initialised (call sxt_init first)");
  return *g_state;
}

// devices[0] of the GPU backend (where single-device helpers stage their data); nothing to hold on
// the host backend
api_state::device_lease lease_primary(api_state& st) {
  return st.backend == SXT_GPU_BACKEND ? st.lease(st.primary()) : api_state::device_lease{};
}

// built-in generators: raw p3 copies on the host (served by sxt_ristretto255_get_generators, feed
// the one-commit chain) and resident addends on EVERY device the backend drives
void init_host_generators(api_state& st, u64 n) {
  st.host_generators.resize(n);
  st.host_one_commits.resize(n);
  if (n == 0) return;
  if (st.backend == SXT_GPU_BACKEND) {
    // derive on the device (reference K15)
    for (auto& dsp : st.devices) {
      device_state& ds = *dsp;
      ds.activate();
      ed_point* d_raw = nullptr;
      BZ_HIP_CHECK(hipMalloc(&d_raw, sizeof(ed_point) * n));
      builtin_generators_enqueue(d_raw, 0, n, ds.stream);
      g_kernel_launches += 1;
      ds.builtin.build(curve25519_vtable(), d_raw, false, n, ds.stream);
      if (ds.slot == 0) {
        BZ_HIP_CHECK(hipMemcpyAsync(st.host_generators.data(), d_raw, sizeof(ed_point) * n,
                                    hipMemcpyDeviceToHost, ds.stream));
      }
      BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
      BZ_HIP_CHECK(hipFree(d_raw));
    }
    st.primary().activate();
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
    device_state& ds = st.primary();
    ds.activate();
    ds.io.reset(sizeof(ed_point) * rest + 256, ds.stream);
    ed_point* d = ds.io.take<ed_point>(rest);
    builtin_generators_enqueue(d, first, rest, ds.stream);
    g_kernel_launches += 1;
    BZ_HIP_CHECK(hipMemcpyAsync(out + done, d, sizeof(ed_point) * rest, hipMemcpyDeviceToHost,
                                ds.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
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
// sxt/multiexp/pippenger/exponent_aggregates_computation.cc:99-101).  The engine indexes the rows of
// a pass with 31 bits (32-bit sorted entries = row | sign << 31); sequences of any u64 length
// (sxt/multiexp/base/exponent_sequence.h:25-42) are cut into passes of at most
// g_max_rows_per_pass rows whose projective partial results are folded (exact group addition).
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
    r.cols[i] = byte_column(d.data, d.n, d.element_nbytes, d.is_signed != 0);
    r.longest = std::max<u64>(r.longest, d.n);
    r.total_bytes += device_arena::padded(static_cast<size_t>(d.n) * d.element_nbytes + 32);
  }
  return r;
}

enum class generator_source { host_api, builtin };

struct generator_ref {
  generator_source source;
  const void* host_generators; // host_api: C-ABI layout, first generator of the range
  u64 offset;                  // builtin: index of the first generator
  // host_api generators this device already holds as a resident set (BLITZAR_AMD_GENERATOR_CACHE,
  // api/state.h): nothing of them is uploaded or converted
  const resident_table* cached = nullptr;
};

// hash of a 1-in-256 sample of the rows (every 256th and the last), whole rows, 8 bytes at a time
u64 sample_hash_of(const void* generators, u64 n, size_t row_bytes) {
  const u8* base = static_cast<const u8*>(generators);
  u64 h = 0x9e3779b97f4a7c15ull ^ n;
  auto mix_row = [&](u64 i) {
    const u8* row = base + i * row_bytes;
    for (size_t k = 0; k + 8 <= row_bytes; k += 8) {
      u64 w;
      std::memcpy(&w, row + k, 8);
      h = (h ^ w) * 0xff51afd7ed558ccdull;
      h ^= h >> 29;
    }
  };
  for (u64 i = 0; i < n; i += 256) mix_row(i);
  if (n != 0) mix_row(n - 1);
  return h;
}

constexpr u64 kGeneratorCacheMinRows = u64{1} << 14;

// The resident set for the caller's host generators on this device, or nullptr (first sightings,
// short sets, knob off).  Called with the device's lease held.
const resident_table* cached_caller_generators(api_state& st, device_state& ds,
                                               const curve_vtable& vt, const void* generators,
                                               u64 n) {
  if (!st.generator_cache || generators == nullptr || n < kGeneratorCacheMinRows) return nullptr;
  const u64 hash = sample_hash_of(generators, n, vt.api_generator_size);
  generator_cache_entry* hit = nullptr;
  generator_cache_entry* victim = &ds.caller_cache[0];
  for (auto& e : ds.caller_cache) {
    if (e.host == generators && e.n == n && e.curve == vt.curve_id && e.sample_hash == hash) hit = &e;
    if (e.last_use < victim->last_use) victim = &e;
  }
  if (hit == nullptr) {
    // a new key (or the same pointer with other content): remember it, drop what the slot held
    ds.activate();
    BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
    victim->table.release();
    *victim = generator_cache_entry{};
    victim->host = generators;
    victim->n = n;
    victim->curve = vt.curve_id;
    victim->sample_hash = hash;
    victim->sightings = 1;
    victim->last_use = ++ds.cache_clock;
    return nullptr;
  }
  hit->last_use = ++ds.cache_clock;
  hit->sightings += 1;
  if (!hit->built) {
    // second sighting: register the set (one more upload of the generators, once)
    ds.activate();
    void* d_tmp = nullptr;
    const size_t bytes = vt.api_generator_size * n;
    BZ_HIP_CHECK(hipMalloc(&d_tmp, bytes));
    BZ_HIP_CHECK(hipMemcpyAsync(d_tmp, generators, bytes, hipMemcpyHostToDevice, ds.stream));
    hit->table.build(vt, d_tmp, false, n, ds.stream);
    BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
    BZ_HIP_CHECK(hipFree(d_tmp));
    hit->built = true;
    st.cache_builds.fetch_add(1);
  } else {
    st.cache_hits.fetch_add(1);
  }
  return &hit->table;
}

//--------------------------------------------------------------------------------------------------
// multi-device plumbing
//--------------------------------------------------------------------------------------------------
struct unit_range {
  size_t begin, end;
};

// contiguous split of weighted units into `parts` ranges of near-equal weight (some may be empty)
std::vector<unit_range> split_by_weight(const std::vector<double>& weight, size_t parts) {
  std::vector<unit_range> r(parts);
  double total = 0;
  for (double w : weight) total += w;
  size_t at = 0;
  double prefix = 0;
  for (size_t p = 0; p < parts; ++p) {
    r[p].begin = at;
    const double goal = total * static_cast<double>(p + 1) / static_cast<double>(parts);
    while (at < weight.size() && (p + 1 == parts || prefix + weight[at] / 2 < goal)) {
      prefix += weight[at];
      ++at;
    }
    r[p].end = at;
  }
  return r;
}

// work of a column in bucket additions (rows x windows), the unit the engine's time follows
std::vector<double> column_weights(const std::vector<host_column>& cols) {
  std::vector<double> weight(cols.size());
  for (size_t i = 0; i < cols.size(); ++i) {
    weight[i] = static_cast<double>(cols[i].n) * ((cols[i].bit_width + 15) / 16) + 1.0;
  }
  return weight;
}

// a row split gives every shard at least 1024 rows
size_t row_split_parts(size_t shards, u64 longest) {
  return static_cast<size_t>(std::min<u64>(shards, std::max<u64>(1, longest / 1024)));
}

// rows [row_begin, row_end) of every column (columns shorter than the range keep what they have)
std::vector<host_column> row_range_of(const std::vector<host_column>& cols, u64 row_begin,
                                      u64 row_end) {
  std::vector<host_column> mine = cols;
  for (auto& c : mine) {
    const u64 b = std::min<u64>(row_begin, c.n), e = std::min<u64>(row_end, c.n);
    c.data = c.data == nullptr ? nullptr : c.data + b * c.row_stride;
    c.n = e - b;
  }
  return mine;
}

// fn(k) for k < count, fn(0) on the calling thread and every other k on its own host thread (each
// thread makes its device current itself); returns when all have finished
template <class F> void run_on_devices(size_t count, F&& fn) {
  std::vector<std::thread> workers;
  workers.reserve(count);
  for (size_t k = 1; k < count; ++k) workers.emplace_back([&fn, k] { fn(k); });
  fn(0);
  for (auto& t : workers) t.join();
}

// bytes of scalars below which a call stays on one device (threads + extra synchronisations cost
// ~0.1 ms); tests lower it to force the sharded paths on small inputs
std::atomic<u64> g_shard_min_bytes{u64{1} << 20};

// rows of a sequence one pass of the engine takes (2^28: 77 GB of generators + addends per pass on
// curve25519); longer sequences run in several passes.  Tests lower it to force the multi-pass path.
std::atomic<u64> g_max_rows_per_pass{u64{1} << 28};

// Enqueue the commitment of `cols` (HOST column pointers) on one device: stage operands into the
// device's io arena, run the engine, leave the results in the arena.  Returns the device pointer of
// the `cols.size()` results (`out_stride` apart); nothing is synchronised.  The columns are
// uploaded in chunks on a copy stream while the engine works on the previous chunk (the DMA
// engines and the CUs are independent): the reference benchmark's 10 x 2^20 x 32-byte job spends a
// third of its time in H2D copies otherwise.
u8* enqueue_commitments(api_state& st, device_state& ds, const curve_vtable& vt,
                        std::vector<host_column> cols, u64 longest, const generator_ref& gens,
                        u32 out_stride, bool projective_out, std::vector<hipEvent_t>& events) {
  ds.activate();
  size_t total_bytes = 0;
  for (const auto& c : cols) {
    total_bytes += device_arena::padded(static_cast<size_t>(c.n) * c.row_stride + 32);
  }
  const size_t gen_bytes = gens.cached != nullptr ? 0
                           : gens.source == generator_source::host_api
                               ? device_arena::padded(vt.api_generator_size * longest + 32) +
                                     device_arena::padded(vt.addend_size * (longest + 1))
                               : device_arena::padded(sizeof(ed_point) * (longest + 1)) +
                                     device_arena::padded(vt.addend_size * (longest + 1));
  const size_t out_bytes = device_arena::padded(static_cast<size_t>(out_stride) * cols.size());
  ds.io.reset(total_bytes + gen_bytes + out_bytes + 1024, ds.stream);
  if (ds.copy_stream == nullptr) {
    BZ_HIP_CHECK(hipStreamCreateWithFlags(&ds.copy_stream, hipStreamNonBlocking));
  }
  auto signal = [&](hipStream_t from, hipStream_t to) {
    hipEvent_t e;
    BZ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BZ_HIP_CHECK(hipEventRecord(e, from));
    BZ_HIP_CHECK(hipStreamWaitEvent(to, e, 0));
    events.push_back(e);
  };
  // the io arena may have been reallocated on ds.stream: order the copy stream behind it
  signal(ds.stream, ds.copy_stream);

  const void* d_addends = nullptr;
  bool resident = false;
  window_table tables{};
  if (gens.cached != nullptr) {
    d_addends = gens.cached->rows_from(0, vt.resident_addend_size);
    resident = true;
    if (gens.cached->tables() != nullptr && longest <= gens.cached->shape.stride) {
      tables = gens.cached->shape;
    }
  } else if (gens.source == generator_source::host_api) {
    u8* d_api = ds.io.take<u8>(vt.api_generator_size * longest + 32);
    void* prepared = ds.io.take<u8>(vt.addend_size * (longest + 1));
    d_addends = prepared;
    if (longest > 0) {
      BZ_HIP_CHECK(hipMemcpyAsync(d_api, gens.host_generators, vt.api_generator_size * longest,
                                  hipMemcpyHostToDevice, ds.copy_stream));
      signal(ds.copy_stream, ds.stream);
      // many columns over these generators: their window table, built once for all column chunks
      // of the call (msm/plan.h, choose_call_table), instead of the plain per-call addends
      const void* table = vt.call_table(*ds.ctx, cols, d_api, &tables, ds.stream);
      if (table != nullptr) {
        d_addends = table;
        resident = true;
      } else {
        vt.prepare_addends(prepared, d_api, longest, ds.stream);
        g_kernel_launches += 1;
      }
    }
  } else if (gens.offset <= st.host_generators.size() &&
             longest <= st.host_generators.size() - gens.offset &&
             ds.builtin.d_addends != nullptr) {
    d_addends = ds.builtin.rows_from(gens.offset, vt.resident_addend_size);
    resident = true;
    // the window-table slices stay `stride` apart; rows of a slice past the set's end are only
    // ever paired with zero digits
    if (ds.builtin.tables() != nullptr && longest <= ds.builtin.shape.stride - gens.offset) {
      tables = ds.builtin.shape;
    }
  } else {
    // beyond the init-time cache: derived on the fly for any offset, like the reference
    // (precomputed_generators.cc:56-91)
    ed_point* d_raw = ds.io.take<ed_point>(longest + 1);
    void* prepared = ds.io.take<u8>(vt.addend_size * (longest + 1));
    builtin_generators_enqueue(d_raw, gens.offset, longest, ds.stream);
    vt.prepare_addends(prepared, d_raw, longest, ds.stream);
    g_kernel_launches += 2;
    d_addends = prepared;
  }
  u8* d_out = ds.io.take<u8>(static_cast<size_t>(out_stride) * cols.size());

  constexpr size_t kChunkBytes = size_t{48} << 20;
  for (size_t begin = 0; begin < cols.size();) {
    size_t end = begin, bytes_in_chunk = 0;
    while (end < cols.size() && (end == begin || bytes_in_chunk < kChunkBytes)) {
      host_column& col = cols[end];
      if (col.n == 0) {
        col.data = nullptr;
      } else {
        const size_t bytes = static_cast<size_t>(col.n) * col.row_stride;
        u8* d = ds.io.take<u8>(bytes + 32);
        BZ_HIP_CHECK(hipMemcpyAsync(d, col.data, bytes, hipMemcpyHostToDevice, ds.copy_stream));
        col.data = d;
        bytes_in_chunk += bytes;
      }
      ++end;
    }
    signal(ds.copy_stream, ds.stream);
    const std::vector<host_column> chunk(cols.begin() + begin, cols.begin() + end);
    u8* out_k = d_out + begin * static_cast<size_t>(out_stride);
    // throughput mode between the chunks of a call (msm_context::tail): the bucket reduction and
    // the Horner chain of a chunk of a few long columns run beside the next chunk (a call of one
    // chunk has nothing to overlap with and keeps the plain path)
    if (begin != 0 || end < cols.size()) msm_context_defer_next_tail(ds.ctx);
    if (resident) {
      vt.msm_resident(*ds.ctx, out_k, out_stride, projective_out, chunk, d_addends, ds.stream,
                      tables.windows != 0 ? &tables : nullptr);
    } else {
      vt.msm(*ds.ctx, out_k, out_stride, projective_out, chunk, d_addends, nullptr, ds.stream);
    }
    begin = end;
  }
  msm_context_join_tail(ds.ctx, ds.stream);
  return d_out;
}

// rows of a blocking call in a pipeline (below): 0 = the cost model decides, k = k row chunks
// wherever the longest column has at least k rows (tests)
std::atomic<u32> g_row_pipeline_chunks{0};

// How many row chunks a single-device call with host operands is cut into, and how many of its
// columns take part in them (`lead`; the others follow as whole columns).  The upload of a call runs
// at the link's rate whatever we do (measured on the MI355X boxes: hipMemcpyAsync from pageable memory
// 56.5 GB/s, from pinned memory 57.5 -- profiles/round4_h2d_rates.txt), so what a call can gain is
// overlap: chunk k computes while chunk k + 1 uploads.  Chunks cost work -- every chunk of every
// column reduces its own buckets and runs its own Horner chain -- so only as many columns are cut
// as it takes to keep the device busy while the caller's generators arrive.
struct row_pipeline_shape {
  u32 chunks = 1;
  u32 lead = 0;
};
row_pipeline_shape choose_row_chunks(const curve_vtable& vt, const std::vector<host_column>& cols,
                                     u64 longest, bool uploads_generators) {
  const u32 num_cols = static_cast<u32>(cols.size());
  const u32 forced = g_row_pipeline_chunks.load();
  if (forced != 0) {
    return {static_cast<u32>(std::min<u64>(forced, std::max<u64>(longest, 1))),
            uploads_generators ? std::min(4u, num_cols) : num_cols};
  }
  // Measured on MI355X (profiles/round4_hostapi_chunks*.txt, round4_hostapi_timeline_*.txt;
  // curve25519, 2^20 rows): the pipeline pays where the caller's generators dominate the upload --
  // 160 bytes of generator against 32 bytes of scalar per row and column -- and nothing of the call
  // could start before they are all there.  The first four columns are committed chunk by chunk
  // while the generators stream in (a chunk's converted addends stay: one full set is resident when
  // the last chunk is through), the other columns follow whole, overlapping their own uploads as in
  // enqueue_commitments.  One column of 2^20 rows: 4.86 ms unpipelined, 4.65 in 8 chunks; two: 6.38 ->
  // 5.45; ten: 14.46 unpipelined, 13.97 / 14.00 / 13.28 / 13.44 / 14.80 with 2 / 3 / 4 / 6 / 10 lead
  // columns (every chunk of every lead column reduces its own buckets: more of them cost more than the
  // idle device they fill).  A call on resident generators uploads too little for chunks to recover
  // what they cost (8 chunks: 1.78 -> 2.96 ms).
  (void)vt;
  if (!uploads_generators || longest < (u64{1} << 19)) return {1, 0};
  return {8, std::min(4u, num_cols)};
}

// The same commitments as enqueue_commitments, as a pipeline over `chunks` row ranges for the first
// `lead` columns: a host thread uploads range k + 1 -- the caller's generators of the range, then the
// lead columns' rows of it -- while the engine converts the generators of range k into their slice of
// ONE full addend array and commits the lead columns' rows of range k to projective partials
// (throughput mode: the tails of range k run beside the front of range k + 1).  The columns after the
// lead follow whole on the complete addend array, their uploads overlapping the computation before
// them; one fold kernel adds the lead columns' partials up and encodes.  Group addition is exact: the
// commitments are the same bytes.
// (ONE upload thread and copy stream: with the generators and the rows on a thread and a stream each
// the link has no idle gaps, but every memory-bound kernel that runs beside TWO copies in flight takes
// 3-4x as long -- k_recode 9 -> 50 us, k_group_scatter 26 -> 109, k_prepare_addends 15 -> 113 -- and a
// one-column call went from 4.65 to 7.2 ms: profiles/round4_hostapi_timeline_*_two_upload_threads.txt.)
u8* enqueue_commitments_row_pipeline(api_state& st, device_state& ds, const curve_vtable& vt,
                                     const std::vector<host_column>& cols, u64 longest,
                                     const generator_ref& gens, u32 out_stride, bool projective_out,
                                     row_pipeline_shape shape, std::vector<hipEvent_t>& events) {
  ds.activate();
  const bool upload_generators = gens.source == generator_source::host_api;
  const u32 num_sequences = static_cast<u32>(cols.size());
  const u32 chunks = shape.chunks;
  const u32 lead = std::max(1u, std::min(shape.lead, num_sequences));
  const std::vector<host_column> lead_cols(cols.begin(), cols.begin() + lead);
  std::vector<host_column> rest_cols(cols.begin() + lead, cols.end());
  const u32 psize = static_cast<u32>(vt.projective_size);
  const u64 rows_max = (longest + chunks - 1) / chunks + 1;
  // staging regions in rotation: the upload of chunk k waits for the computation of chunk
  // k - kRegions only (with two regions it kept running into the computation of chunk k - 2, which
  // shares the device with the upload of chunk k - 1)
  constexpr u32 kRegions = 3;
  // one staging region: the caller's generators of a range + the lead columns' rows of it
  size_t region_bytes = 0;
  if (upload_generators) region_bytes += device_arena::padded(vt.api_generator_size * rows_max + 32);
  for (const auto& c : lead_cols) {
    region_bytes += device_arena::padded(static_cast<size_t>(std::min<u64>(c.n, rows_max)) * c.row_stride + 32);
  }
  size_t rest_bytes = 0;
  for (const auto& c : rest_cols) {
    rest_bytes += device_arena::padded(static_cast<size_t>(c.n) * c.row_stride + 32);
  }
  const size_t addend_bytes =
      upload_generators ? device_arena::padded(vt.addend_size * (longest + 1)) : 0;
  const size_t partial_bytes = static_cast<size_t>(psize) * lead;
  ds.io.reset(kRegions * region_bytes + addend_bytes + rest_bytes +
                  device_arena::padded(partial_bytes * chunks) +
                  device_arena::padded(static_cast<size_t>(out_stride) * num_sequences) + 4096,
              ds.stream);
  if (ds.copy_stream == nullptr) {
    BZ_HIP_CHECK(hipStreamCreateWithFlags(&ds.copy_stream, hipStreamNonBlocking));
  }
  auto new_event = [&] {
    hipEvent_t e;
    BZ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    events.push_back(e);
    return e;
  };
  struct region {
    u8* api = nullptr;
    std::vector<u8*> columns;
  } regions[kRegions];
  for (auto& r : regions) {
    if (upload_generators) r.api = ds.io.take<u8>(vt.api_generator_size * rows_max + 32);
    for (const auto& c : lead_cols) {
      r.columns.push_back(ds.io.take<u8>(static_cast<size_t>(std::min<u64>(c.n, rows_max)) * c.row_stride + 32));
    }
  }
  u8* d_addends_all = upload_generators ? ds.io.take<u8>(vt.addend_size * (longest + 1)) : nullptr;
  std::vector<u8*> d_rest;
  for (const auto& c : rest_cols) {
    d_rest.push_back(c.n == 0 ? nullptr : ds.io.take<u8>(static_cast<size_t>(c.n) * c.row_stride + 32));
  }
  u8* d_partials = ds.io.take<u8>(partial_bytes * chunks);
  u8* d_out = ds.io.take<u8>(static_cast<size_t>(out_stride) * num_sequences);
  // the io arena may have been reallocated on ds.stream: the copy streams start behind it
  hipEvent_t ready = new_event();
  BZ_HIP_CHECK(hipEventRecord(ready, ds.stream));
  BZ_HIP_CHECK(hipStreamWaitEvent(ds.copy_stream, ready, 0));
  const bool resident = !upload_generators;
  window_table tables{};
  if (resident && ds.builtin.tables() != nullptr && longest <= ds.builtin.shape.stride - gens.offset) {
    tables = ds.builtin.shape;
  }
  // the columns after the lead in batches of whole columns, as enqueue_commitments cuts them
  constexpr size_t kBatchBytes = size_t{48} << 20;
  std::vector<std::pair<size_t, size_t>> batches;
  for (size_t begin = 0; begin < rest_cols.size();) {
    size_t end = begin, bytes = 0;
    while (end < rest_cols.size() && (end == begin || bytes < kBatchBytes)) {
      bytes += static_cast<size_t>(rest_cols[end].n) * rest_cols[end].row_stride;
      ++end;
    }
    batches.emplace_back(begin, end);
    begin = end;
  }
  std::vector<hipEvent_t> computed(chunks), copied(chunks), copied_batch(batches.size());
  for (u32 k = 0; k < chunks; ++k) {
    computed[k] = new_event();
    copied[k] = new_event();
  }
  for (auto& e : copied_batch) e = new_event();
  struct chunk_range {
    u64 begin, end;
    std::vector<host_column> columns; // the lead columns' rows of the range (host pointers)
  };
  std::vector<chunk_range> ranges(chunks);
  for (u32 k = 0; k < chunks; ++k) {
    ranges[k].begin = static_cast<u64>(static_cast<unsigned __int128>(longest) * k / chunks);
    ranges[k].end = static_cast<u64>(static_cast<unsigned __int128>(longest) * (k + 1) / chunks);
    ranges[k].columns = row_range_of(lead_cols, ranges[k].begin, ranges[k].end);
  }
  std::mutex mu;
  std::condition_variable cv;
  u32 issued = 0;         // chunks whose copies (and `copied` event) are in the copy stream
  u32 issued_batches = 0; // the same for the batches of whole columns
  u32 enqueued = 0;          // chunks whose `computed` event is in the compute stream
  const int device = ds.device;
  auto publish = [&](u32& counter, u32 value) {
    {
      std::lock_guard<std::mutex> lock(mu);
      counter = value;
    }
    cv.notify_all();
  };
  // hipMemcpyAsync from pageable memory occupies the calling host thread for the length of the copy
  // (the runtime stages the data itself), and enqueueing a chunk's kernels costs ~0.1 ms of host
  // time: done by one thread, every enqueue is a gap in the upload (measured: 8 chunks 4.99 ms
  // against 4.86 unpipelined).  So the uploads run on a helper thread, back to back; the calling
  // thread enqueues chunk k's work as soon as the helper has issued chunk k's copies.
  std::thread uploader([&] {
    BZ_HIP_CHECK(hipSetDevice(device));
    for (u32 k = 0; k < chunks; ++k) {
      region& r = regions[k % kRegions];
      if (k >= kRegions) {
        // the region is free once chunk k - kRegions has been computed: wait until that event has
        // been recorded by the other thread, then let the copy stream wait for it
        {
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return enqueued >= k - kRegions + 1; });
        }
        BZ_HIP_CHECK(hipStreamWaitEvent(ds.copy_stream, computed[k - kRegions], 0));
      }
      const chunk_range& cr = ranges[k];
      if (upload_generators && cr.end > cr.begin) {
        BZ_HIP_CHECK(hipMemcpyAsync(r.api, static_cast<const u8*>(gens.host_generators) +
                                               vt.api_generator_size * cr.begin,
                                    vt.api_generator_size * (cr.end - cr.begin),
                                    hipMemcpyHostToDevice, ds.copy_stream));
      }
      for (size_t c = 0; c < cr.columns.size(); ++c) {
        if (cr.columns[c].n == 0) continue;
        BZ_HIP_CHECK(hipMemcpyAsync(r.columns[c], cr.columns[c].data,
                                    static_cast<size_t>(cr.columns[c].n) * cr.columns[c].row_stride,
                                    hipMemcpyHostToDevice, ds.copy_stream));
      }
      BZ_HIP_CHECK(hipEventRecord(copied[k], ds.copy_stream));
      publish(issued, k + 1);
    }
    for (size_t b = 0; b < batches.size(); ++b) {
      for (size_t c = batches[b].first; c < batches[b].second; ++c) {
        if (rest_cols[c].n == 0) continue;
        BZ_HIP_CHECK(hipMemcpyAsync(d_rest[c], rest_cols[c].data,
                                    static_cast<size_t>(rest_cols[c].n) * rest_cols[c].row_stride,
                                    hipMemcpyHostToDevice, ds.copy_stream));
      }
      BZ_HIP_CHECK(hipEventRecord(copied_batch[b], ds.copy_stream));
      publish(issued_batches, static_cast<u32>(b + 1));
    }
  });
  const bool several_calls = chunks > 1 || !batches.empty();
  for (u32 k = 0; k < chunks; ++k) {
    region& r = regions[k % kRegions];
    {
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return issued > k; });
    }
    BZ_HIP_CHECK(hipStreamWaitEvent(ds.stream, copied[k], 0));
    std::vector<host_column> mine = ranges[k].columns;
    for (size_t c = 0; c < mine.size(); ++c) mine[c].data = mine[c].n == 0 ? nullptr : r.columns[c];
    const u64 rows = ranges[k].end - ranges[k].begin;
    u8* out_k = d_partials + partial_bytes * k;
    if (several_calls) msm_context_defer_next_tail(ds.ctx);
    if (upload_generators) {
      u8* addends_k = d_addends_all + vt.addend_size * ranges[k].begin;
      if (rows > 0) {
        vt.prepare_addends(addends_k, r.api, rows, ds.stream);
        g_kernel_launches += 1;
      }
      vt.msm(*ds.ctx, out_k, psize, true, mine, addends_k, nullptr, ds.stream);
    } else {
      // a range of a resident set: the same rows of every window-table slice
      const void* d_addends =
          ds.builtin.rows_from(gens.offset + ranges[k].begin, vt.resident_addend_size);
      vt.msm_resident(*ds.ctx, out_k, psize, true, mine, d_addends, ds.stream,
                      tables.windows != 0 ? &tables : nullptr);
    }
    BZ_HIP_CHECK(hipEventRecord(computed[k], ds.stream));
    publish(enqueued, k + 1);
  }
  // the other columns, whole: every addend is in place behind the last chunk's conversion
  for (size_t b = 0; b < batches.size(); ++b) {
    {
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return issued_batches > b; });
    }
    BZ_HIP_CHECK(hipStreamWaitEvent(ds.stream, copied_batch[b], 0));
    std::vector<host_column> batch(rest_cols.begin() + batches[b].first,
                                   rest_cols.begin() + batches[b].second);
    for (size_t c = 0; c < batch.size(); ++c) {
      batch[c].data = batch[c].n == 0 ? nullptr : d_rest[batches[b].first + c];
    }
    u8* out_b = d_out + (lead + batches[b].first) * static_cast<size_t>(out_stride);
    msm_context_defer_next_tail(ds.ctx);
    if (upload_generators) {
      vt.msm(*ds.ctx, out_b, out_stride, projective_out, batch, d_addends_all, nullptr, ds.stream);
    } else {
      vt.msm_resident(*ds.ctx, out_b, out_stride, projective_out, batch,
                      ds.builtin.rows_from(gens.offset, vt.resident_addend_size), ds.stream,
                      tables.windows != 0 ? &tables : nullptr);
    }
  }
  uploader.join();
  msm_context_join_tail(ds.ctx, ds.stream);
  if (projective_out) {
    vt.fold_device(d_out, d_partials, chunks, lead, ds.stream);
  } else {
    vt.fold_encode_device(d_out, d_partials, chunks, lead, ds.stream);
  }
  g_kernel_launches += 1;
  (void)st;
  return d_out;
}

// does a call with these columns spread over all devices of the GPU backend?
bool shards_over_devices(const api_state& st, const checked_columns& cc) {
  size_t scalar_bytes = 0;
  for (const auto& c : cc.cols) scalar_bytes += static_cast<size_t>(c.n) * c.row_stride;
  return st.devices.size() > 1 && scalar_bytes >= g_shard_min_bytes.load() && cc.longest > 0;
}

// The Pedersen path of all five entry points.  GPU backend: the caller holds the lease of `single`
// (the call stays on that device) or, with `single` == nullptr, of every device (the call may
// shard).  Sequences of more than one pass fold their partials on devices[0]: `single` must be that
// device then (compute_commitments sees to it).
void compute_commitments_locked(api_state& st, const curve_vtable& vt, void* commitments,
                                u32 num_sequences, const sxt_sequence_descriptor* descriptors,
                                const void* generators, generator_source source,
                                u64 offset_generators, bool projective_out,
                                device_state* single) {
  checked_columns cc = check_descriptors(descriptors, num_sequences);
  const u32 out_stride = static_cast<u32>(projective_out ? vt.projective_size : vt.output_size);

  size_t scalar_bytes = 0;
  for (const auto& c : cc.cols) scalar_bytes += static_cast<size_t>(c.n) * c.row_stride;
  u8* out = static_cast<u8*>(commitments);

  if (st.backend == SXT_CPU_BACKEND) {
    std::vector<ed_point> builtin;
    const u8* gens = static_cast<const u8*>(generators);
    if (source == generator_source::builtin) {
      builtin.resize(cc.longest);
      host_builtin_generators(st, builtin.data(), cc.longest, offset_generators);
      gens = reinterpret_cast<const u8*>(builtin.data());
    }
    // BLITZAR_AMD_FORCE_SHARDS on the host backend: the same split rules as the GPU backend below,
    // one host thread per shard (GPU-less coverage of the sharding logic; the reference cpu
    // backend is single-threaded)
    const size_t shards = st.host_shards;
    if (shards < 2 || projective_out || scalar_bytes < g_shard_min_bytes.load() ||
        cc.longest == 0) {
      vt.msm_host(out, out_stride, projective_out, cc.cols, gens, false, cc.longest);
    } else if (num_sequences >= shards) {
      const std::vector<unit_range> ranges = split_by_weight(column_weights(cc.cols), shards);
      run_on_devices(shards, [&](size_t k) {
        const unit_range r = ranges[k];
        if (r.begin == r.end) return;
        const std::vector<host_column> mine(cc.cols.begin() + r.begin, cc.cols.begin() + r.end);
        u64 longest = 0;
        for (const auto& c : mine) longest = std::max<u64>(longest, c.n);
        vt.msm_host(out + r.begin * static_cast<size_t>(out_stride), out_stride, false, mine, gens,
                    false, longest);
      });
    } else {
      const size_t parts = row_split_parts(shards, cc.longest);
      const size_t psize = vt.projective_size;
      std::vector<u8> partials(psize * num_sequences * parts);
      run_on_devices(parts, [&](size_t k) {
        const u64 row_begin = cc.longest * k / parts, row_end = cc.longest * (k + 1) / parts;
        const std::vector<host_column> mine = row_range_of(cc.cols, row_begin, row_end);
        vt.msm_host(partials.data() + psize * num_sequences * k, static_cast<u32>(psize), true,
                    mine, gens + vt.api_generator_size * row_begin, false, row_end - row_begin);
      });
      vt.fold_encode_host(out, partials.data(), static_cast<u32>(parts), num_sequences);
    }
    return;
  }

  // GPU backend.  One device: stage, run, copy back.  Several devices:
  //   * at least as many columns as devices: contiguous column ranges of near-equal work per
  //     device, every device needs the generators of its longest column; the encodings go straight
  //     from each device to the caller's (host) array -- no inter-device traffic at all;
  //   * fewer columns (one long column): every device takes a row range of every column and the
  //     matching generator slice, produces projective partials, the partials travel to device 0
  //     over xGMI (hipMemcpyPeerAsync, <= 160 B per column and device), device 0 folds and encodes
  //     (k_fold_encode).  Group addition is exact, so the canonical result is the same.
  const size_t num_devices = st.devices.size();
  const bool shard = single == nullptr && shards_over_devices(st, cc);
  const generator_ref all_gens{source, generators, offset_generators};
  // sequences longer than one pass of the engine: row ranges, like the row split below
  const u64 max_rows = g_max_rows_per_pass.load();
  const size_t passes = static_cast<size_t>((cc.longest + max_rows - 1) / max_rows);

  if (!shard && passes <= 1) {
    device_state& ds = single != nullptr ? *single : st.primary();
    std::vector<hipEvent_t> events;
    // operands that have to travel: the scalars, and the caller's generators; built-in generators
    // are resident when the call lies inside the init-time cache
    const bool cached_builtin = source == generator_source::builtin &&
                                offset_generators <= st.host_generators.size() &&
                                cc.longest <= st.host_generators.size() - offset_generators &&
                                ds.builtin.d_addends != nullptr;
    generator_ref call_gens = all_gens;
    if (source == generator_source::host_api) {
      call_gens.cached = cached_caller_generators(st, ds, vt, generators, cc.longest);
    }
    row_pipeline_shape shape;
    if (call_gens.cached == nullptr && (source == generator_source::host_api || cached_builtin)) {
      shape = choose_row_chunks(vt, cc.cols, cc.longest, source == generator_source::host_api);
    }
    u8* d_out = shape.chunks > 1
                    ? enqueue_commitments_row_pipeline(st, ds, vt, cc.cols, cc.longest, call_gens,
                                                       out_stride, projective_out, shape, events)
                    : enqueue_commitments(st, ds, vt, cc.cols, cc.longest, call_gens, out_stride,
                                          projective_out, events);
    BZ_HIP_CHECK(hipMemcpyAsync(out, d_out, static_cast<size_t>(out_stride) * num_sequences,
                                hipMemcpyDeviceToHost, ds.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
    for (auto& e : events) (void)hipEventDestroy(e);
    return;
  }

  // everything below works through devices[0]'s staging, context and gather buffers: the caller
  // holds that device's lease (lease_all, or lease(primary) for a call in several passes).  The
  // limits are test knobs re-read here; if one moved between the caller's choice of lease and this
  // point, stop instead of racing on a device that is not ours.
  BZ_RELEASE_ASSERT(single == nullptr || single == &st.primary(),
                    "bzamd_set_max_rows_per_pass / bzamd_set_shard_min_bytes changed during a call");
  if (shard && num_sequences >= num_devices && passes <= 1) {
    const std::vector<unit_range> ranges = split_by_weight(column_weights(cc.cols), num_devices);
    run_on_devices(num_devices, [&](size_t k) {
      const unit_range r = ranges[k];
      if (r.begin == r.end) return;
      device_state& ds = *st.devices[k];
      std::vector<host_column> mine(cc.cols.begin() + r.begin, cc.cols.begin() + r.end);
      u64 longest = 0;
      for (const auto& c : mine) longest = std::max<u64>(longest, c.n);
      std::vector<hipEvent_t> events;
      u8* d_out = enqueue_commitments(st, ds, vt, std::move(mine), longest, all_gens, out_stride,
                                      projective_out, events);
      BZ_HIP_CHECK(hipMemcpyAsync(out + r.begin * static_cast<size_t>(out_stride), d_out,
                                  static_cast<size_t>(out_stride) * (r.end - r.begin),
                                  hipMemcpyDeviceToHost, ds.stream));
      BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
      for (auto& e : events) (void)hipEventDestroy(e);
    });
    st.primary().activate();
    return;
  }

  // row split: over the devices, and / or into passes of at most max_rows rows (a device takes
  // every workers-th part, one after the other)
  const u32 psize = static_cast<u32>(vt.projective_size);
  const size_t parts = std::max(row_split_parts(shard ? num_devices : 1, cc.longest), passes);
  const size_t workers = shard ? std::min(num_devices, parts) : 1;
  device_state& root = st.primary();
  root.activate();
  const size_t partial_bytes = static_cast<size_t>(psize) * num_sequences;
  st.gather.reset(partial_bytes * parts + device_arena::padded(out_stride * num_sequences) + 512,
                  root.stream);
  u8* d_partials = st.gather.take<u8>(partial_bytes * parts);
  u8* d_final = st.gather.take<u8>(static_cast<size_t>(out_stride) * num_sequences);
  BZ_HIP_CHECK(hipStreamSynchronize(root.stream)); // the gather buffer exists before peers write
  run_on_devices(workers, [&](size_t w) {
    device_state& ds = *st.devices[w];
    for (size_t k = w; k < parts; k += workers) {
      const u64 row_begin = static_cast<u64>(static_cast<unsigned __int128>(cc.longest) * k / parts);
      const u64 row_end = static_cast<u64>(static_cast<unsigned __int128>(cc.longest) * (k + 1) / parts);
      std::vector<host_column> mine = row_range_of(cc.cols, row_begin, row_end);
      generator_ref g = all_gens;
      if (g.source == generator_source::host_api) {
        g.host_generators = static_cast<const u8*>(generators) + vt.api_generator_size * row_begin;
      } else {
        g.offset += row_begin;
      }
      std::vector<hipEvent_t> events;
      u8* d_part = enqueue_commitments(st, ds, vt, std::move(mine), row_end - row_begin, g, psize,
                                       true, events);
      BZ_HIP_CHECK(hipMemcpyPeerAsync(d_partials + partial_bytes * k, root.device, d_part,
                                      ds.device, partial_bytes, ds.stream));
      BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
      for (auto& e : events) (void)hipEventDestroy(e);
    }
  });
  root.activate();
  if (projective_out) {
    vt.fold_device(d_final, d_partials, static_cast<u32>(parts), num_sequences, root.stream);
  } else {
    vt.fold_encode_device(d_final, d_partials, static_cast<u32>(parts), num_sequences,
                          root.stream);
  }
  g_kernel_launches += 1;
  BZ_HIP_CHECK(hipMemcpyAsync(out, d_final, static_cast<size_t>(out_stride) * num_sequences,
                              hipMemcpyDeviceToHost, root.stream));
  BZ_HIP_CHECK(hipStreamSynchronize(root.stream));
}

// a positive count from the environment, `fallback` when unset; anything else aborts
long env_count(const char* name, long fallback) {
  const char* val = std::getenv(name);
  if (val == nullptr || val[0] == 0) return fallback;
  char* end = nullptr;
  const long v = std::strtol(val, &end, 10);
  BZ_RELEASE_ASSERT(end != nullptr && *end == 0 && v >= 1 && v <= 64,
                    "BLITZAR_AMD_NUM_DEVICES / BLITZAR_AMD_FORCE_SHARDS must be in [1, 64]");
  return v;
}

void compute_commitments(const curve_vtable& vt, void* commitments, u32 num_sequences,
                         const sxt_sequence_descriptor* descriptors, const void* generators,
                         generator_source source, u64 offset_generators,
                         bool projective_out = false) {
  if (num_sequences == 0) return; // reference: returns before touching anything
  BZ_RELEASE_ASSERT(commitments != nullptr, "commitments is null");
  api_state& st = state();
  // (one line as the call starts, one as it completes: bucket_method2/multiexponentiation.h:55,71)
  struct call_log {
    u32 outputs;
    u64 longest = 0;
    call_log(u32 num, const sxt_sequence_descriptor* d) : outputs{num} {
      if (!log::enabled(log::info) || d == nullptr) return;
      for (u32 i = 0; i < num; ++i) longest = d[i].n > longest ? d[i].n : longest;
      BZ_LOG_INFO("compute a multiexponentiation with %u outputs of length %llu", outputs,
                  static_cast<unsigned long long>(longest));
    }
    ~call_log() {
      BZ_LOG_INFO("finished multiexponentiation with %u outputs of length %llu", outputs,
                  static_cast<unsigned long long>(longest));
    }
  } logged{num_sequences, descriptors};
  if (st.backend != SXT_GPU_BACKEND) { // the host backend keeps no per-call state: no lock
    compute_commitments_locked(st, vt, commitments, num_sequences, descriptors, generators, source,
                               offset_generators, projective_out, nullptr);
    return;
  }
  // which devices the call needs is a function of its shapes alone
  const checked_columns cc = check_descriptors(descriptors, num_sequences);
  const bool several_passes = cc.longest > g_max_rows_per_pass.load();
  const current_device_guard restore_callers_device; // the call may run on any device (lease_any)
  if (shards_over_devices(st, cc)) {
    api_state::device_lease lease = st.lease_all();
    compute_commitments_locked(st, vt, commitments, num_sequences, descriptors, generators, source,
                               offset_generators, projective_out, nullptr);
  } else {
    api_state::device_lease lease = several_passes ? st.lease(st.primary()) : st.lease_any();
    compute_commitments_locked(st, vt, commitments, num_sequences, descriptors, generators, source,
                               offset_generators, projective_out, lease.device);
  }
}

int backend_from_environment(int backend) {
  const char* val = std::getenv("BLITZAR_BACKEND");
  if (val == nullptr) return backend;
  std::string s{val};
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  BZ_LOG_INFO("override default backend with environmental variable BLITZAR_BACKEND=%s", s.c_str());
  if (s == "cpu") return SXT_CPU_BACKEND;
  if (s == "gpu") return SXT_GPU_BACKEND;
  std::fprintf(stderr, "blitzar_amd: invalid BLITZAR_BACKEND value %s\n", val);
  std::abort();
}
} // namespace

api_state* current_state() { return g_state; }

namespace proof {
void host_builtin_generators_unlocked(api_state& st, ed_point* out, u64 n, u64 offset) {
  host_builtin_generators(st, out, n, offset);
}
void commit_column_unlocked(api_state& st, u8* out32, const u8* scalars, u64 n,
                            const ed_point* generators) {
  sxt_sequence_descriptor d{};
  d.element_nbytes = 32;
  d.n = n;
  d.data = scalars;
  d.is_signed = 0;
  // (the proof entry points hold the lease of devices[0])
  compute_commitments_locked(st, curve25519_vtable(), out32, 1, &d, generators,
                             generator_source::host_api, 0, false,
                             st.backend == SXT_GPU_BACKEND ? &st.primary() : nullptr);
}
} // namespace proof
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
  BZ_LOG_INFO(backend == SXT_GPU_BACKEND ? "initializing GPU backend" : "initializing CPU backend");
  if (backend == SXT_GPU_BACKEND) {
    // no silent fallback: a GPU backend without a GPU is a hard error, as in the reference
    // (cbindings/backend.cc:61-63 "no supported GPUs found")
    const int visible = device_count();
    BZ_RELEASE_ASSERT(visible > 0, "no supported GPUs found");
    int current = 0;
    BZ_HIP_CHECK(hipGetDevice(&current));
    // devices this backend drives (api/state.h): the current one first, then the others
    std::vector<int> ids;
    const long forced = env_count("BLITZAR_AMD_FORCE_SHARDS", 0);
    if (forced > 1) {
      ids.assign(static_cast<size_t>(forced), current);
    } else {
      const long cap = env_count("BLITZAR_AMD_NUM_DEVICES", visible);
      ids.push_back(current);
      for (int d = 0; d < visible && static_cast<long>(ids.size()) < cap; ++d) {
        if (d != current) ids.push_back(d);
      }
    }
    for (size_t k = 0; k < ids.size(); ++k) {
      auto ds = std::make_unique<device_state>();
      ds->slot = static_cast<int>(k);
      ds->device = ids[k];
      ds->activate();
      BZ_HIP_CHECK(hipStreamCreateWithFlags(&ds->stream, hipStreamNonBlocking));
      ds->ctx = msm_context_new();
      if (ids[k] != current) {
        // results of a row-split call travel to device 0 with hipMemcpyPeerAsync
        int can = 0;
        BZ_HIP_CHECK(hipDeviceCanAccessPeer(&can, ids[k], current));
        if (can) {
          const hipError_t e = hipDeviceEnablePeerAccess(current, 0);
          if (e != hipSuccess) (void)hipGetLastError(); // already enabled
        }
      }
      st->devices.push_back(std::move(ds));
    }
    BZ_HIP_CHECK(hipSetDevice(current));
  } else {
    st->host_shards = static_cast<size_t>(env_count("BLITZAR_AMD_FORCE_SHARDS", 1));
  }
  if (const char* v = std::getenv("BLITZAR_AMD_GENERATOR_CACHE")) {
    st->generator_cache = !(v[0] == '0' && v[1] == 0) && v[0] != 0;
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
  const api_state::device_lease lease = lease_primary(st);
  host_builtin_generators(st, reinterpret_cast<ed_point*>(generators), num_generators,
                          offset_generators);
  return 0;
}

int sxt_curve25519_get_one_commit(struct sxt_ristretto255* one_commit, uint64_t n) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(one_commit != nullptr, "one_commit is null");
  const api_state::device_lease lease = lease_primary(st);
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
  if (val == nullptr) {
    BZ_LOG_INFO("using a default partition window width of %u", 16u);
    return 16;
  }
  const long w = std::strtol(val, nullptr, 10);
  BZ_RELEASE_ASSERT(w > 0 && w <= 32, "invalid BLITZAR_PARTITION_WINDOW_WIDTH");
  return static_cast<unsigned>(w);
}

void handle_make_resident(multiexp_handle& h) {
  api_state& st = state();
  if (st.backend != SXT_GPU_BACKEND || h.n == 0) return;
  const size_t bytes = h.vt->projective_size * h.n;
  const current_device_guard restore_callers_device;
  for (auto& dsp : st.devices) {
    device_state& ds = *dsp;
    const api_state::device_lease lease = st.lease(ds);
    ds.activate();
    void* d_proj = nullptr;
    BZ_HIP_CHECK(hipMalloc(&d_proj, bytes));
    BZ_HIP_CHECK(hipMemcpyAsync(d_proj, h.host_projective.data(), bytes, hipMemcpyHostToDevice,
                                ds.stream));
    resident_table table;
    table.build(*h.vt, d_proj, true, h.n, ds.stream);
    BZ_HIP_CHECK(hipFree(d_proj));
    h.tables.push_back(table);
    h.devices.push_back(ds.device);
  }
}

// the three fixed-base entry points differ only in how a row is cut into per-output bit fields
// 
void fixed_multiexponentiation(void* res, const multiexp_handle& h, const unsigned* bit_table,
                               unsigned uniform_bits, const unsigned* lengths,
                               unsigned num_outputs, unsigned n, const u8* scalars,
                               bool device_operands, hipStream_t caller_stream,
                               bool record = true) {
  if (num_outputs == 0) return;
  BZ_RELEASE_ASSERT(res != nullptr, "res is null");
  api_state& st = state();
  if (log::enabled(log::info) && record) {
    // (pippenger2/multiexponentiation.h:254,261)
    unsigned long long products = 0;
    for (unsigned k = 0; k < num_outputs; ++k) products += bit_table != nullptr ? bit_table[k] : uniform_bits;
    BZ_LOG_INFO("computing %llu bitwise multiexponentiation products of length %u", products, n);
  }
  struct done_log {
    unsigned outputs;
    bool on;
    ~done_log() {
      if (on) BZ_LOG_INFO("completed %u reductions", outputs);
    }
  } done{num_outputs, record};
  // An output wider than 256 bits (the reference takes any unsigned width,
  // cbindings/blitzar_api.h:712, pippenger2/multiexponentiation.h:207-288: a bit plane per bit): the
  // engine's columns are scalars of at most 256 bits, so such an output is computed as
  // ceil(width / 256) adjacent columns of the same rows and folded, sum_j 2^(256 j) piece_j, with the
  // curve's own doubling and addition (exact group arithmetic: the same point)
  bool any_wide = false;
  for (unsigned k = 0; k < num_outputs; ++k) {
    any_wide = any_wide || (bit_table != nullptr ? bit_table[k] : uniform_bits) > 256;
  }
  if (any_wide) {
    BZ_RELEASE_ASSERT(!device_operands,
                      "device entry point: outputs wider than 256 bits are not supported");
    std::vector<unsigned> piece_bits, piece_lengths;
    std::vector<u32> piece_counts(num_outputs);
    for (unsigned k = 0; k < num_outputs; ++k) {
      unsigned width = bit_table != nullptr ? bit_table[k] : uniform_bits;
      BZ_RELEASE_ASSERT(width > 0, "output bit width must be positive");
      piece_counts[k] = (width + 255) / 256;
      for (; width > 0; width -= std::min(width, 256u)) {
        piece_bits.push_back(std::min(width, 256u));
        piece_lengths.push_back(lengths != nullptr ? lengths[k] : n);
      }
    }
    BZ_RELEASE_ASSERT(piece_bits.size() < (size_t{1} << 31), "too many output pieces");
    const size_t psize = h.vt->projective_size;
    std::vector<u8> pieces(psize * piece_bits.size());
    // BLITZAR_DUMP_DIR records the call as the caller made it
    std::unique_ptr<dump_recorder> recorder;
    u64 total_bits = 0;
    unsigned max_len = 0;
    for (size_t i = 0; i < piece_bits.size(); ++i) {
      total_bits += piece_bits[i];
      max_len = std::max(max_len, piece_lengths[i]);
    }
    if (bit_table != nullptr && record) {
      recorder = std::make_unique<dump_recorder>(lengths != nullptr ? "vlen-multiexponentiation"
                                                                     : "packed-multiexponentiation");
      if (recorder->recording()) {
        recorder->write_inputs(h, bit_table, lengths, num_outputs, max_len, scalars,
                               static_cast<size_t>((total_bits + 7) / 8) * max_len);
      }
    }
    fixed_multiexponentiation(pieces.data(), h, piece_bits.data(), 0,
                              lengths != nullptr ? piece_lengths.data() : nullptr,
                              static_cast<unsigned>(piece_bits.size()), n, scalars, false, nullptr,
                              false);
    h.vt->fold_shifted_host(static_cast<u8*>(res), pieces.data(), piece_counts.data(), num_outputs,
                            256);
    if (recorder && recorder->recording()) {
      recorder->write("result.bin", res, psize * num_outputs);
    }
    return;
  }
  u64 total_bits = 0;
  unsigned prev_len = 0, max_len = 0;
  std::vector<host_column> cols(num_outputs);
  for (unsigned k = 0; k < num_outputs; ++k) {
    const unsigned width = bit_table != nullptr ? bit_table[k] : uniform_bits;
    BZ_RELEASE_ASSERT(width > 0, "output bit width must be positive");
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
  // (a handle of 2^31 generators would need 275 GB of resident addends: bounded by memory, not by
  // the interface)
  BZ_RELEASE_ASSERT(max_len < (1u << 31), "fixed-base calls are limited to 2^31 - 1 rows");
  BZ_RELEASE_ASSERT(max_len == 0 || scalars != nullptr, "scalars is null");
  std::vector<u64> first_bit(num_outputs);
  for (unsigned k = 0; k < num_outputs; ++k) {
    host_column& c = cols[k];
    first_bit[k] = c.bit_offset;
    c.row_stride = row_bytes;
    // fold whole bytes of the bit offset into the base pointer
    c.data = scalars + (c.bit_offset >> 3);
    c.bit_offset &= 7;
  }
  const u32 out_stride = static_cast<u32>(h.vt->projective_size);

  if (device_operands) {
    // asynchronous on the caller's stream and device: no shared staging, no api lock
    BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
    int dev = 0;
    BZ_HIP_CHECK(hipGetDevice(&dev));
    const resident_table* table = h.table_on(dev);
    BZ_RELEASE_ASSERT(table != nullptr, "the handle has no addends on the current device");
    apply_pipeline_request(st.context_for_current_device());
    h.vt->msm_resident(*st.context_for_current_device(), static_cast<u8*>(res), out_stride, true,
                       cols, table->d_addends, caller_stream, table->tables());
    return;
  }

  // BLITZAR_DUMP_DIR: record packed / vlen calls with host operands (the plain byte-aligned entry
  // point is not recorded by the reference either, gpu_backend.cc:257-272)
  std::unique_ptr<dump_recorder> recorder;
  if (bit_table != nullptr && record) {
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

  // outputs are independent units -- contiguous output ranges of near-equal work per device
  // (SURVEY 8(e)); BLITZAR_AMD_FORCE_SHARDS applies the same split to the host backend's threads
  const size_t num_devices =
      st.backend == SXT_CPU_BACKEND ? st.host_shards : st.devices.size();
  const size_t scalar_bytes = static_cast<size_t>(row_bytes) * max_len;
  const bool shard = num_devices > 1 && num_outputs >= num_devices && max_len > 0 &&
                     scalar_bytes >= g_shard_min_bytes.load();
  std::vector<unit_range> ranges{{0, num_outputs}};
  if (shard) ranges = split_by_weight(column_weights(cols), num_devices);
  u8* out = static_cast<u8*>(res);
  // GPU backend: every device for a sharded call, else the first one nobody holds; the calling
  // thread gets its own current device back when the call returns
  api_state::device_lease lease;
  std::unique_ptr<current_device_guard> restore_callers_device;
  if (st.backend == SXT_GPU_BACKEND) {
    restore_callers_device = std::make_unique<current_device_guard>();
    lease = shard ? st.lease_all() : st.lease_any();
  }

  if (st.backend == SXT_CPU_BACKEND) {
    run_on_devices(ranges.size(), [&](size_t k) {
      const unit_range r = ranges[k];
      if (r.begin == r.end) return;
      const std::vector<host_column> mine(cols.begin() + r.begin, cols.begin() + r.end);
      u64 rows = 0;
      for (const auto& c : mine) rows = std::max<u64>(rows, c.n);
      h.vt->msm_host(out + r.begin * static_cast<size_t>(out_stride), out_stride, true, mine,
                     h.host_projective.data(), true, rows);
    });
    record_result();
    return;
  }

  // GPU backend: the outputs of a call are bit fields of the same rows, so a device uploads only
  // the byte span of every row that its outputs occupy (strided 2-D copy)
  run_on_devices(ranges.size(), [&](size_t k) {
    const unit_range r = ranges[k];
    if (r.begin == r.end) return;
    device_state& ds = shard ? *st.devices[k] : *lease.device;
    ds.activate();
    std::vector<host_column> mine(cols.begin() + r.begin, cols.begin() + r.end);
    u64 rows = 0;
    for (const auto& c : mine) rows = std::max<u64>(rows, c.n);
    const size_t outputs = r.end - r.begin;
    const size_t out_bytes = static_cast<size_t>(out_stride) * outputs;
    // bytes [span_begin, span_end) of every row hold this range's bit fields
    const u64 span_begin = first_bit[r.begin] >> 3;
    const u64 last_bit = first_bit[r.end - 1] + cols[r.end - 1].bit_width;
    const u64 span = (last_bit + 7) / 8 - span_begin;
    ds.io.reset(device_arena::padded(span * rows + 64) + device_arena::padded(out_bytes) + 512,
                ds.stream);
    u8* d_scalars = ds.io.take<u8>(span * rows + 64);
    if (rows > 0 && span == row_bytes) {
      BZ_HIP_CHECK(hipMemcpyAsync(d_scalars, scalars, span * rows, hipMemcpyHostToDevice,
                                  ds.stream));
    } else if (rows > 0) {
      BZ_HIP_CHECK(hipMemcpy2DAsync(d_scalars, span, scalars + span_begin, row_bytes, span, rows,
                                    hipMemcpyHostToDevice, ds.stream));
    }
    for (auto& c : mine) {
      c.data = d_scalars + ((c.data - scalars) - span_begin);
      c.row_stride = span;
    }
    u8* d_out = ds.io.take<u8>(out_bytes);
    h.vt->msm_resident(*ds.ctx, d_out, out_stride, true, mine, h.tables[ds.slot].d_addends,
                       ds.stream, h.tables[ds.slot].tables());
    BZ_HIP_CHECK(hipMemcpyAsync(out + r.begin * static_cast<size_t>(out_stride), d_out, out_bytes,
                                hipMemcpyDeviceToHost, ds.stream));
    BZ_HIP_CHECK(hipStreamSynchronize(ds.stream));
  });
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
  // GPU backend: the 2^w subset sums of every window are built on the device
  // (fixed/partition_table_device.h); wider windows than 16 bits and the cpu backend use the
  // reference's serial host recurrence
  api_state& st = state();
  bool ok = false;
  if (st.backend == SXT_GPU_BACKEND && h->window_width <= 16 && h->n > 0) {
    const api_state::device_lease lease = st.lease(st.primary());
    device_state& ds = st.primary();
    ds.activate();
    ok = h->vt->write_partition_table_device(f, h->window_width, h->host_projective.data(), h->n,
                                             ds.stream);
  } else {
    ok = h->vt->write_partition_table(f, h->window_width, h->host_projective.data(), h->n);
  }
  // a full disk must not leave a silently truncated table behind
  BZ_RELEASE_ASSERT(std::fclose(f) == 0 && ok, "short write to the partition table file");
}

void sxt_multiexp_handle_free(struct sxt_multiexp_handle* handle) {
  auto* h = reinterpret_cast<multiexp_handle*>(handle);
  if (h == nullptr) return;
  if (!h->tables.empty()) {
    int current = 0;
    (void)hipGetDevice(&current);
    for (size_t k = 0; k < h->tables.size(); ++k) {
      (void)hipSetDevice(h->devices[k]);
      h->tables[k].release();
    }
    (void)hipSetDevice(current);
  }
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
// inner-product argument (proof/inner_product.hip) and sumcheck prover (proof/sumcheck.hip)
//--------------------------------------------------------------------------------------------------
void sxt_curve25519_prove_inner_product(struct sxt_ristretto255_compressed* l_vector,
                                        struct sxt_ristretto255_compressed* r_vector,
                                        struct sxt_curve25519_scalar* ap_value,
                                        struct sxt_transcript* transcript, uint64_t n,
                                        uint64_t generators_offset,
                                        const struct sxt_curve25519_scalar* a_vector,
                                        const struct sxt_curve25519_scalar* b_vector) {
  // validation: cbindings/inner_product_proof.cc:35-57
  BZ_RELEASE_ASSERT(transcript != nullptr, "transcript must not be null");
  BZ_RELEASE_ASSERT(ap_value != nullptr, "ap_value must not be null");
  BZ_RELEASE_ASSERT(b_vector != nullptr, "b_vector must not be null");
  BZ_RELEASE_ASSERT(a_vector != nullptr, "a_vector must not be null");
  BZ_RELEASE_ASSERT(n > 0, "a_vector and b_vector lengths must be greater than zero");
  BZ_RELEASE_ASSERT(n == 1 || (l_vector != nullptr && r_vector != nullptr),
                    "l_vector and r_vector must not be null when n is bigger than one");
  BZ_RELEASE_ASSERT(n <= (uint64_t{1} << 30), "inner products are limited to 2^30 elements");
  api_state& st = state();
  const api_state::device_lease lease = lease_primary(st);
  proof::prove_inner_product(st, reinterpret_cast<u8*>(l_vector), reinterpret_cast<u8*>(r_vector),
                             ap_value->bytes, transcript, n, generators_offset,
                             reinterpret_cast<const u8*>(a_vector),
                             reinterpret_cast<const u8*>(b_vector));
}

int sxt_curve25519_verify_inner_product(struct sxt_transcript* transcript, uint64_t n,
                                        uint64_t generators_offset,
                                        const struct sxt_curve25519_scalar* b_vector,
                                        const struct sxt_curve25519_scalar* product,
                                        const struct sxt_ristretto255* a_commit,
                                        const struct sxt_ristretto255_compressed* l_vector,
                                        const struct sxt_ristretto255_compressed* r_vector,
                                        const struct sxt_curve25519_scalar* ap_value) {
  // validation: cbindings/inner_product_proof.cc:62-91 (aborts on null inputs even though the
  // proof itself is untrusted, like the reference)
  BZ_RELEASE_ASSERT(transcript != nullptr, "transcript must not be null");
  BZ_RELEASE_ASSERT(ap_value != nullptr, "ap_value must not be null");
  BZ_RELEASE_ASSERT(product != nullptr, "product must not be null");
  BZ_RELEASE_ASSERT(a_commit != nullptr, "a_commit must not be null");
  BZ_RELEASE_ASSERT(b_vector != nullptr, "b_vector must not be null");
  BZ_RELEASE_ASSERT(n > 0, "b_vector length must be greater than zero");
  BZ_RELEASE_ASSERT(n == 1 || (l_vector != nullptr && r_vector != nullptr),
                    "l_vector and r_vector must not be null when n is bigger than one");
  BZ_RELEASE_ASSERT(n <= (uint64_t{1} << 30), "inner products are limited to 2^30 elements");
  api_state& st = state();
  const api_state::device_lease lease = lease_primary(st);
  return proof::verify_inner_product(st, transcript, n, generators_offset,
                                     reinterpret_cast<const u8*>(b_vector), product->bytes,
                                     a_commit, reinterpret_cast<const u8*>(l_vector),
                                     reinterpret_cast<const u8*>(r_vector), ap_value->bytes)
             ? 1
             : 0;
}

void sxt_prove_sumcheck(void* polynomials, void* evaluation_point, unsigned field_id,
                        const struct sumcheck_descriptor* descriptor, void* transcript_callback,
                        void* transcript_context) {
  BZ_RELEASE_ASSERT(polynomials != nullptr && evaluation_point != nullptr && descriptor != nullptr &&
                        transcript_callback != nullptr,
                    "null argument to `sxt_prove_sumcheck`");
  BZ_RELEASE_ASSERT(descriptor->mles != nullptr && descriptor->product_table != nullptr &&
                        descriptor->product_terms != nullptr,
                    "null table in the sumcheck descriptor");
  api_state& st = state();
  // devices[0] is held while the prover works on it and given up around every call of the caller's
  // transcript callback, which may therefore call back into this library (the tables of the proof
  // live in memory of the call's own, not in the device's staging arena)
  api_state::device_lease lease = lease_primary(st);
  const proof::sumcheck_inputs in{descriptor->mles,         descriptor->product_table,
                                  descriptor->product_terms, descriptor->n,
                                  descriptor->num_mles,      descriptor->num_products,
                                  descriptor->num_product_terms, descriptor->round_degree};
  proof::prove_sumcheck(st, polynomials, evaluation_point, field_id, in, transcript_callback,
                        transcript_context, &lease);
}

//--------------------------------------------------------------------------------------------------
// extensions (include/blitzar_amd.h)
//--------------------------------------------------------------------------------------------------
const char* bzamd_version(void) { return "blitzar_amd 0.1 (gfx950)"; }

int bzamd_device_count(void) { return device_count(); }

int bzamd_active_backend(void) { return g_state == nullptr ? 0 : g_state->backend; }

int bzamd_num_devices(void) {
  if (g_state == nullptr) return 0;
  return static_cast<int>(g_state->backend == SXT_GPU_BACKEND ? g_state->devices.size()
                                                              : g_state->host_shards);
}

void bzamd_set_shard_min_bytes(uint64_t bytes) { g_shard_min_bytes.store(bytes); }

void bzamd_set_row_pipeline_chunks(uint32_t chunks) {
  BZ_RELEASE_ASSERT(chunks <= 64, "row pipeline: at most 64 chunks (0 = automatic)");
  g_row_pipeline_chunks.store(chunks);
}

void bzamd_set_max_rows_per_pass(uint64_t rows) {
  BZ_RELEASE_ASSERT(rows >= 1 && rows < (uint64_t{1} << 31), "rows per pass must be in [1, 2^31)");
  g_max_rows_per_pass.store(rows);
}

void bzamd_transcript_init(struct sxt_transcript* transcript, const char* label,
                           uint64_t label_len) {
  BZ_RELEASE_ASSERT(transcript != nullptr && (label != nullptr || label_len == 0), "null argument");
  proof::transcript::init(transcript, std::string_view{label, label_len});
}

int bzamd_accumulate_form(void) {
  // curve25519 caller generators keep their projective form (msm/msm_curve25519.hip: the per-call
  // normalisation was measured and rejected); resident sets use the Z = 1 form
  return 0;
}

uint64_t bzamd_kernel_launch_count(void) { return g_kernel_launches.load(); }

void bzamd_generator_cache_stats(uint64_t* hits, uint64_t* builds) {
  if (hits != nullptr) *hits = g_state == nullptr ? 0 : g_state->cache_hits.load();
  if (builds != nullptr) *builds = g_state == nullptr ? 0 : g_state->cache_builds.load();
}

int bzamd_probe_mad_rate(double target_ms, double* out) {
  if (g_state == nullptr || g_state->backend != SXT_GPU_BACKEND || out == nullptr) return -1;
  api_state& st = *g_state;
  // the probe loads every SIMD of the CURRENT device for `target_ms`: if the backend drives that
  // device, hold its lease so that no blocking sxt_* call is timed (or slowed) beside it
  int device = 0;
  BZ_HIP_CHECK(hipGetDevice(&device));
  api_state::device_lease lease;
  for (auto& d : st.devices) {
    if (d->device == device) {
      lease = st.lease(*d);
      break;
    }
  }
  return msm_probe_mad_rate(target_ms, out) ? 0 : -1;
}

int bzamd_slow_instruction_fetch(void) {
  if (g_state == nullptr || g_state->backend != SXT_GPU_BACKEND) return -1;
  return msm_context_slow_instruction_fetch(g_state->context_for_current_device()) ? 1 : 0;
}

uint32_t bzamd_concurrent_calls_high_water(void) {
  return g_state == nullptr ? 0 : g_state->in_flight_high.load();
}

void bzamd_stage_timing_begin(uint64_t max_calls) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "stage timing needs the GPU backend");
  msm_context_timing_begin(st.context_for_current_device(), max_calls);
}

void bzamd_stage_timing_begin_masked(uint64_t max_calls, uint32_t stage_mask) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "stage timing needs the GPU backend");
  msm_context_timing_begin(st.context_for_current_device(), max_calls, stage_mask);
}

void bzamd_stage_timing_begin_sampled(uint64_t max_calls, uint32_t stage_mask, uint32_t sample_every) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "stage timing needs the GPU backend");
  msm_context_timing_begin(st.context_for_current_device(), max_calls, stage_mask, sample_every);
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

void bzamd_set_window_bits(uint32_t window_bits) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "tuning applies to the GPU backend");
  msm_context_set_window_bits(st.context_for_current_device(), window_bits);
  for (auto& d : st.devices) msm_context_set_window_bits(d->ctx, window_bits);
}

uint64_t bzamd_set_call_tables(int mode) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "tuning applies to the GPU backend");
  uint64_t built = msm_context_set_call_tables(st.context_for_current_device(), mode);
  for (auto& d : st.devices) {
    if (d->ctx != st.context_for_current_device()) built += msm_context_set_call_tables(d->ctx, mode);
  }
  return built;
}

void bzamd_set_segments(uint32_t log2_entries_per_accumulate_lane,
                        uint32_t log2_buckets_per_reduce_lane) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "tuning applies to the GPU backend");
  msm_context_set_segments(st.context_for_current_device(), log2_entries_per_accumulate_lane,
                           log2_buckets_per_reduce_lane);
}

void bzamd_reset_for_testing(void) {
  if (g_state == nullptr) return;
  delete g_state;
  g_state = nullptr;
}

namespace {
// Device operands, sequences longer than one pass of the engine (g_max_rows_per_pass): every pass
// commits a row range of all columns to projective partials, one fold kernel adds them up (and
// encodes).  `d_addends` (resident set, `addend_size` bytes per row) or `d_api_generators`.
void msm_device_in_passes(const curve_vtable& vt, msm_context* ctx, u8* out, bool projective_out,
                          const checked_columns& cc, const void* d_addends, size_t addend_size,
                          const void* d_api_generators, hipStream_t stream) {
  const u64 max_rows = g_max_rows_per_pass.load();
  const size_t passes = static_cast<size_t>((cc.longest + max_rows - 1) / max_rows);
  const u32 psize = static_cast<u32>(vt.projective_size);
  const u32 num_sequences = static_cast<u32>(cc.cols.size());
  const size_t partial_bytes = static_cast<size_t>(psize) * num_sequences;
  u8* d_partials = nullptr;
  BZ_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&d_partials), partial_bytes * passes, stream));
  for (size_t k = 0; k < passes; ++k) {
    const u64 row_begin = static_cast<u64>(static_cast<unsigned __int128>(cc.longest) * k / passes);
    const u64 row_end = static_cast<u64>(static_cast<unsigned __int128>(cc.longest) * (k + 1) / passes);
    const std::vector<host_column> mine = row_range_of(cc.cols, row_begin, row_end);
    if (d_addends != nullptr) {
      vt.msm_resident(*ctx, d_partials + partial_bytes * k, psize, true, mine,
                      static_cast<const u8*>(d_addends) + addend_size * row_begin, stream, nullptr);
    } else {
      vt.msm(*ctx, d_partials + partial_bytes * k, psize, true, mine, nullptr,
             static_cast<const u8*>(d_api_generators) + vt.api_generator_size * row_begin, stream);
    }
  }
  if (projective_out) {
    vt.fold_device(out, d_partials, static_cast<u32>(passes), num_sequences, stream);
  } else {
    vt.fold_encode_device(out, d_partials, static_cast<u32>(passes), num_sequences, stream);
  }
  g_kernel_launches += 1;
  BZ_HIP_CHECK(hipFreeAsync(d_partials, stream));
}

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
  if (cc.longest > g_max_rows_per_pass.load()) {
    t_pipeline_next = false; // a call of several passes completes on the caller's stream
    msm_device_in_passes(*vt, st.context_for_current_device(), static_cast<u8*>(out),
                         projective_out, cc, nullptr, 0, generators,
                         static_cast<hipStream_t>(stream));
    return;
  }
  apply_pipeline_request(st.context_for_current_device());
  vt->msm(*st.context_for_current_device(), static_cast<u8*>(out),
          static_cast<u32>(projective_out ? vt->projective_size : vt->output_size), projective_out,
          cc.cols, nullptr, generators, static_cast<hipStream_t>(stream));
}
} // namespace

void bzamd_pipeline_next(void) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  t_pipeline_next = true;
}

void bzamd_pipeline_flush(void* stream) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  msm_context_join_tail(st.context_for_current_device(), static_cast<hipStream_t>(stream));
}

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

//--------------------------------------------------------------------------------------------------
// multi-device MSM inside one process, device-resident operands, RCCL all-gather of the results
//--------------------------------------------------------------------------------------------------
int bzamd_device_id(int slot) {
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  BZ_RELEASE_ASSERT(slot >= 0 && static_cast<size_t>(slot) < st.devices.size(), "no such device slot");
  return st.devices[static_cast<size_t>(slot)]->device;
}

uint32_t bzamd_multi_device_columns_per_device(uint32_t num_sequences) {
  api_state& st = state();
  const uint32_t d = static_cast<uint32_t>(st.devices.empty() ? 1 : st.devices.size());
  return (num_sequences + d - 1) / d;
}

const char* bzamd_multi_device_exchange(void) {
  api_state& st = state();
  return st.exchange_state == 1 ? "rccl" : (st.exchange_state == 2 ? "peer-copies" : "unused");
}

void bzamd_msm_multi_device(unsigned curve_id, void* const* commitments, uint32_t num_sequences,
                            const struct sxt_sequence_descriptor* descriptors,
                            const void* const* generators) {
  if (num_sequences == 0) return;
  const curve_vtable* vt = curve_vtable_for(curve_id);
  BZ_RELEASE_ASSERT(vt != nullptr, "unknown curve id");
  BZ_RELEASE_ASSERT(generators != nullptr, "generators is null");
  api_state& st = state();
  BZ_RELEASE_ASSERT(st.backend == SXT_GPU_BACKEND, "device entry points need the GPU backend");
  const api_state::device_lease lease = st.lease_all();
  const size_t D = st.devices.size();
  const checked_columns cc = check_descriptors(descriptors, num_sequences);
  // (one pass of the engine per column here: its sorted entries hold a row in 31 bits)
  BZ_RELEASE_ASSERT(cc.longest <= g_max_rows_per_pass.load(),
                    "bzamd_msm_multi_device: sequences longer than one engine pass are not supported");
  const u32 out_stride = static_cast<u32>(vt->output_size);
  const size_t per = (num_sequences + D - 1) / D; // columns per device (the last ones may be short)
  const size_t chunk = per * out_stride;          // bytes every device contributes
  int current = 0;
  BZ_HIP_CHECK(hipGetDevice(&current));

  // the exchange: RCCL over the devices' links, unless two slots share a physical device (RCCL
  // refuses duplicates) or there is only one device (nothing to exchange: librccl is never mapped).
  // ncclCommInitAll runs on a helper thread with a deadline (BLITZAR_AMD_RCCL_INIT_TIMEOUT_S,
  // default 60): a first-ever bring-up of the fabric that never returns must not hang the caller --
  // the exchange falls back to peer copies and says so.
  if (st.exchange_state == 0) {
    // (one device: tests still drive the RCCL path -- library load, communicator, collective -- with
    // a single rank through BLITZAR_AMD_RCCL_SINGLE_DEVICE=1)
    const char* single = std::getenv("BLITZAR_AMD_RCCL_SINGLE_DEVICE");
    bool distinct = D > 1 || (single != nullptr && single[0] == '1');
    for (size_t a = 0; a < D; ++a) {
      for (size_t b = a + 1; b < D; ++b) distinct = distinct && st.devices[a]->device != st.devices[b]->device;
    }
    rccl_api* rccl = distinct ? rccl_api::get() : nullptr;
    st.exchange_state = 2;
    if (rccl != nullptr) {
      struct init_job {
        std::vector<int> ids;
        std::vector<ncclComm_t> comms;
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        bool abandoned = false; // the deadline passed: whoever finishes the init cleans up after it
        ncclResult_t result = ncclSuccess;
      };
      auto job = std::make_shared<init_job>();
      job->ids.resize(D);
      job->comms.resize(D);
      for (size_t d = 0; d < D; ++d) job->ids[d] = st.devices[d]->device;
      std::thread([job, rccl] {
        const ncclResult_t r =
            rccl->comm_init_all(job->comms.data(), static_cast<int>(job->ids.size()), job->ids.data());
        bool late = false;
        {
          std::lock_guard<std::mutex> lock(job->mu);
          job->result = r;
          job->done = true;
          late = job->abandoned;
          job->cv.notify_all();
        }
        // finished after the deadline: nobody will ever use (or destroy) these communicators
        if (late && r == ncclSuccess) {
          for (ncclComm_t c : job->comms) (void)rccl->comm_destroy(c);
        }
      }).detach();
      long deadline_s = 60;
      if (const char* v = std::getenv("BLITZAR_AMD_RCCL_INIT_TIMEOUT_S")) {
        const long parsed = std::strtol(v, nullptr, 10);
        if (parsed >= 1 && parsed <= 3600) deadline_s = parsed;
      }
      std::unique_lock<std::mutex> lock(job->mu);
      const bool in_time =
          job->cv.wait_for(lock, std::chrono::seconds(deadline_s), [&] { return job->done; });
      if (in_time && job->result == ncclSuccess) {
        for (auto c : job->comms) st.comms.push_back(c);
        st.destroy_comm = [](void* c) {
          if (rccl_api* api = rccl_api::get()) (void)api->comm_destroy(static_cast<ncclComm_t>(c));
        };
        st.exchange_state = 1;
      } else if (!in_time) {
        // (a hung init leaves its helper thread behind, inside RCCL; if it does return, the thread
        // destroys what it created)
        job->abandoned = true;
        std::fprintf(stderr, "blitzar_amd: ncclCommInitAll did not return within %ld s; using peer "
                             "copies\n", deadline_s);
      } else {
        std::fprintf(stderr, "blitzar_amd: ncclCommInitAll failed (%s); using peer copies\n",
                     rccl->get_error_string != nullptr ? rccl->get_error_string(job->result) : "?");
      }
    }
  }

  // every device: its columns -> its piece of the send buffer (one host thread per device)
  std::vector<u8*> send(D), recv(D);
  run_on_devices(D, [&](size_t d) {
    device_state& ds = *st.devices[d];
    ds.activate();
    ds.io.reset(device_arena::padded(chunk) + device_arena::padded(chunk * D) + 512, ds.stream);
    send[d] = ds.io.take<u8>(chunk);
    recv[d] = ds.io.take<u8>(chunk * D);
    const size_t begin = std::min<size_t>(d * per, num_sequences);
    const size_t end = std::min<size_t>(begin + per, num_sequences);
    if (end - begin < per) BZ_HIP_CHECK(hipMemsetAsync(send[d], 0, chunk, ds.stream));
    if (begin == end) return;
    BZ_RELEASE_ASSERT(generators[d] != nullptr, "generators of a device that owns columns is null");
    const std::vector<host_column> mine(cc.cols.begin() + begin, cc.cols.begin() + end);
    vt->msm(*ds.ctx, send[d], out_stride, false, mine, nullptr, generators[d], ds.stream);
  });

  if (st.exchange_state == 1) {
    rccl_api* rccl = rccl_api::get();
    ncclResult_t r = rccl->group_start();
    for (size_t d = 0; d < D && r == ncclSuccess; ++d) {
      st.devices[d]->activate();
      r = rccl->all_gather(send[d], recv[d], chunk, ncclUint8, static_cast<ncclComm_t>(st.comms[d]),
                           st.devices[d]->stream);
    }
    const ncclResult_t e = rccl->group_end();
    BZ_RELEASE_ASSERT(r == ncclSuccess && e == ncclSuccess, "ncclAllGather failed");
  } else {
    // peer copies: every piece to every device, on the stream of the device that produced it
    // (behind its MSM); the destinations are synchronised below
    for (size_t s = 0; s < D; ++s) {
      device_state& src = *st.devices[s];
      src.activate();
      for (size_t d = 0; d < D; ++d) {
        BZ_HIP_CHECK(hipMemcpyPeerAsync(recv[d] + chunk * s, st.devices[d]->device, send[s],
                                        src.device, chunk, src.stream));
      }
    }
  }
  for (size_t d = 0; d < D; ++d) {
    st.devices[d]->activate();
    BZ_HIP_CHECK(hipStreamSynchronize(st.devices[d]->stream));
  }
  if (commitments != nullptr) {
    for (size_t d = 0; d < D; ++d) {
      if (commitments[d] == nullptr) continue;
      st.devices[d]->activate();
      BZ_HIP_CHECK(hipMemcpyAsync(commitments[d], recv[d], static_cast<size_t>(out_stride) * num_sequences,
                                  hipMemcpyDeviceToDevice, st.devices[d]->stream));
      BZ_HIP_CHECK(hipStreamSynchronize(st.devices[d]->stream));
    }
  }
  BZ_HIP_CHECK(hipSetDevice(current));
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
  g->table.build(*vt, generators, false, n, static_cast<hipStream_t>(stream));
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
  g->table.release();
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
  if (cc.longest > g_max_rows_per_pass.load()) {
    t_pipeline_next = false;
    msm_device_in_passes(*g->vt, st.context_for_current_device(), static_cast<u8*>(commitments),
                         false, cc, g->table.d_addends, g->vt->resident_addend_size, nullptr,
                         static_cast<hipStream_t>(stream));
    return;
  }
  apply_pipeline_request(st.context_for_current_device());
  g->vt->msm_resident(*st.context_for_current_device(), static_cast<u8*>(commitments),
                      static_cast<u32>(g->vt->output_size), false, cc.cols, g->table.d_addends,
                      static_cast<hipStream_t>(stream), g->table.tables());
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
