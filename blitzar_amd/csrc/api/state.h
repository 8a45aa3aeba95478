// Process-wide backend singleton behind the C ABI (reference: the `backend` pointer of
// cbindings/backend.cc:37 plus the leaked generator caches of
// sxt/seqcommit/generator/precomputed_generators.cc:38).
//
// The GPU backend drives every visible HIP device from this one process, like the reference's
// (sxt/execution/device/for_each.cc:56-82 hands chunks to "the next available device",
// gpu_backend.cc:150-193): one `device_state` per device -- its own stream pair, engine context,
// staging arena and replica of the resident built-in generators -- and one host thread per device
// for the duration of a blocking sxt_* call (api/capi.hip).  Knobs, read once at sxt_init:
//   BLITZAR_AMD_NUM_DEVICES   use at most this many devices (default: all visible; bench.py's
//                             one-process-per-GPU ranks set 1)
//   BLITZAR_AMD_FORCE_SHARDS  k > 1: k logical devices on the CURRENT physical device -- the
//                             multi-device code paths (sharding, threads, peer copies, fold) on a
//                             one-GPU box, the way the reference's tests force chunking with
//                             split_options (pippenger2/multiexponentiation.t.cc:150-180)
#pragma once

#include <atomic>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/generators/builtin.h"
#include "blitzar_amd/csrc/msm/dispatch.h"

namespace bz {

// A resident generator set on one device: slice 0 = the set's addends; when window tables are on
// (sets of at least kWindowTableMinGenerators generators, BLITZAR_AMD_WINDOW_TABLES != 0) the
// 2^(bits w) multiples follow, `stride` rows apart -- 17 x the memory at bits = 16 (2.2 GB for 2^20
// curve25519 generators), 15 x at bits = 18 -- which buys one bucket reduction per column instead of
// one per window and no Horner chain.
// Window width of the table.  With ONE bucket set per column more buckets are affordable than with W
// separate sets, and wider windows mean fewer slices to add (256-bit columns: 17 slices at 16 bits,
// 16 at 17, 15 at 18).  Measured on MI355X (profiles/round5_ab_wide_tables_and_short_columns.log, ms
// per call lone / in sequence): bn254, 8 columns x 2^20 rows: 16 bits 11.80 / 10.80, 17 bits 11.05 /
// 10.65, 18 bits 11.63 / 11.19, 19 bits 12.6 / 12.0 -- the additions saved (k_accumulate 9.2 -> 9.0 ms)
// are partly eaten by the sort, whose groups no longer fit the LDS stage once a merged task has 2^17
// buckets over 2^24 virtual rows (0.65 -> 1.35 / 1.67 ms), while k_reduce gains (1.5 -> 0.4 ms: fewer
// head partials per bucket); grumpkin, 64 columns x 2^18 rows: 21.0 / 20.9 / 22.5 ms at 16 / 17 / 18 bits
// and 27.1 at 20; curve25519, 8 x 2^20: 6.2 -> 7.0 ms at 18.  So: 17 bits for Weierstrass sets of 2^20
// generators or more, 16 for everything else.  BLITZAR_AMD_WINDOW_TABLE_BITS (16 .. 20) overrides
// (tests, A/B runs).
constexpr u64 kWindowTableMinGenerators = u64{1} << 14;
constexpr u64 kWideWindowTableMinGenerators = u64{1} << 20;
inline u32 window_table_slices(u32 bits) { return (256 + 1 + bits - 1) / bits; } // + the carry
struct resident_table {
  void* d_addends = nullptr;
  u64 n = 0;
  window_table shape; // shape.windows == 0: plain addends only

  // tables for the rows [offset, offset + ...) of the set, or nullptr
  const window_table* tables() const { return shape.windows != 0 ? &shape : nullptr; }
  const void* rows_from(u64 offset, size_t addend_size) const {
    return static_cast<const char*>(d_addends) + addend_size * offset;
  }
  // build on the current device from C-ABI generators / projective elements already on the device
  void build(const curve_vtable& vt, const void* d_source, bool source_projective, u64 count,
             hipStream_t stream) {
    n = count;
    shape = window_table{};
    const char* env = std::getenv("BLITZAR_AMD_WINDOW_TABLES");
    const bool wanted = env == nullptr || env[0] != '0';
    u64 least = kWindowTableMinGenerators;
    if (const char* v = std::getenv("BLITZAR_AMD_WINDOW_TABLE_MIN")) {
      least = std::strtoull(v, nullptr, 10); // tests: tables for small sets too
    }
    u32 bits = vt.curve_id != 0 && count >= kWideWindowTableMinGenerators ? 17 : 16;
    if (const char* v = std::getenv("BLITZAR_AMD_WINDOW_TABLE_BITS")) {
      const unsigned long b = std::strtoul(v, nullptr, 10);
      BZ_RELEASE_ASSERT(b >= 16 && b <= kMaxTableWindowBits,
                        "BLITZAR_AMD_WINDOW_TABLE_BITS must be in [16, 20]");
      bits = static_cast<u32>(b);
    }
    u32 windows = wanted && count >= least ? window_table_slices(bits) : 1;
    const u64 stride = (count + 7) & ~u64{7};
    const size_t row = vt.resident_addend_size;
    if (windows > 1) {
      // 17 x the memory of the plain addends, on every device the backend drives: only while it is
      // a modest share of what the device has left (a quarter of the free HBM, or
      // BLITZAR_AMD_WINDOW_TABLE_MAX_BYTES) -- a caller who precomputes 2^24 generators and commits
      // short columns would otherwise pay 36 GB per device for tables no call of his uses
      size_t free_bytes = 0, total_bytes = 0;
      BZ_HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
      size_t cap = free_bytes / 4;
      if (const char* v = std::getenv("BLITZAR_AMD_WINDOW_TABLE_MAX_BYTES")) {
        cap = static_cast<size_t>(std::strtoull(v, nullptr, 10));
      }
      if (row * stride * windows > cap) windows = 1;
    }
    hipError_t err = hipMalloc(&d_addends, row * (stride * windows + 1));
    if (err != hipSuccess && windows > 1) {
      (void)hipGetLastError(); // no room for the table after all: plain addends
      windows = 1;
      err = hipMalloc(&d_addends, row * (stride + 1));
    }
    BZ_HIP_CHECK(err);
    vt.build_window_table(d_addends, d_source, source_projective, count, stride, windows, bits,
                          stream);
    if (windows > 1) {
      shape.stride = stride;
      shape.windows = windows;
      shape.bits = bits;
    }
  }
  void release() {
    if (d_addends != nullptr) (void)hipFree(d_addends);
    d_addends = nullptr;
  }
};

// A blocking call may run on any device of the backend (per-device leases) and activates it on the
// calling thread; the thread's own current device -- what its later allocations and bzamd_*_device
// calls resolve against -- is put back when the call returns.
struct current_device_guard {
  int saved = -1;
  current_device_guard() {
    if (hipGetDevice(&saved) != hipSuccess) {
      (void)hipGetLastError();
      saved = -1;
    }
  }
  ~current_device_guard() {
    if (saved >= 0) (void)hipSetDevice(saved);
  }
  current_device_guard(const current_device_guard&) = delete;
  current_device_guard& operator=(const current_device_guard&) = delete;
};

// BLITZAR_AMD_GENERATOR_CACHE=1 (opt-in): a drop-in caller of sxt_*_with_generators hands the same
// host array of generators to call after call (Proof-of-SQL commits many tables against one set) and
// pays its upload every time -- 160 MB of PCIe per call at 2^20 curve25519 generators, what the
// reference does too (sxt/multiexp/bucket_method/accumulation.h:68-71).  With the knob on, a device
// remembers (host pointer, count, curve, hash of a 1-in-256 sample of the rows); the SECOND call
// that shows the same key registers the set as a resident one (its own upload, Z = 1 addends, window
// tables: what bzamd_generators_new_host does), and later calls whose key and sample hash still match
// run on it and upload scalars only.  The caveat that makes it opt-in: a caller who rewrites rows the
// sample misses, in place, under the same pointer, gets commitments to the OLD generators.
struct generator_cache_entry {
  const void* host = nullptr;
  u64 n = 0;
  unsigned curve = 0;
  u64 sample_hash = 0;
  u32 sightings = 0;
  u64 last_use = 0;
  bool built = false;
  resident_table table;
};

struct device_state {
  // A blocking sxt_* call owns the devices it runs on for its duration: their stream pair, engine
  // context and staging arena (api_state::device_lease).  Calls on different devices run
  // concurrently, like the reference's thread_local schedulers and pinned pools
  // (sxt/execution/schedule/scheduler.cc:65-69, sxt/base/device/pinned_buffer_pool.h:52-55).
  std::mutex mu;
  int slot = 0;   // index into api_state::devices (handles keep one addend replica per slot)
  int device = 0; // HIP device id
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr; // H2D of the next chunk of columns beside the computation
  msm_context* ctx = nullptr;
  device_arena io;                   // staging of host operands / results of the blocking calls
  resident_table builtin; // the built-in generators cached at sxt_init
  generator_cache_entry caller_cache[2]; // BLITZAR_AMD_GENERATOR_CACHE (see above)
  u64 cache_clock = 0;

  void activate() const { BZ_HIP_CHECK(hipSetDevice(device)); }
  ~device_state() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    builtin.release();
    for (auto& e : caller_cache) e.table.release();
    if (ctx != nullptr) msm_context_free(ctx);
    if (copy_stream != nullptr) (void)hipStreamDestroy(copy_stream);
    if (stream != nullptr) (void)hipStreamDestroy(stream);
  }
};

struct api_state {
  int backend = 0; // SXT_CPU_BACKEND / SXT_GPU_BACKEND
  // devices[0] is the device that was current at sxt_init: single-device work runs there
  std::vector<std::unique_ptr<device_state>> devices;
  // Locking of the blocking sxt_* entry points: per device, no process-wide lock (round 3 had one).
  //   lease_any()  a call that stays on one device takes the first device nobody holds (slot
  //                order), or queues on them round-robin: two host threads on a two-device backend
  //                run side by side;
  //   lease_all()  a call that shards over the devices takes all of them, in slot order (every
  //                multi-device holder locks in that order: no cycles);
  //   lease(ds)    one specific device.
  // The host backend keeps no per-call state and takes no lock at all.
  struct device_lease {
    std::vector<std::unique_lock<std::mutex>> held;
    device_state* device = nullptr; // lease_any / lease: the device leased
    api_state* owner = nullptr;
    device_lease() = default;
    device_lease(device_lease&&) = default;
    device_lease& operator=(device_lease&&) = default;
    ~device_lease() {
      if (owner != nullptr && !held.empty()) owner->in_flight.fetch_sub(1);
    }
    // give the devices up for the duration of a caller-supplied callback and take them back
    void unlock() {
      for (auto it = held.rbegin(); it != held.rend(); ++it) it->unlock();
    }
    void relock() {
      for (auto& l : held) l.lock();
    }
  };
  std::atomic<u32> next_device{0};
  std::atomic<u32> in_flight{0};        // leases alive now
  std::atomic<u32> in_flight_high{0};   // ... and the most there ever were (tests)
  void note_lease(device_lease& l) {
    l.owner = this;
    const u32 now = in_flight.fetch_add(1) + 1;
    u32 seen = in_flight_high.load();
    while (now > seen && !in_flight_high.compare_exchange_weak(seen, now)) {
    }
  }
  device_lease lease(device_state& ds) {
    device_lease l;
    l.held.emplace_back(ds.mu);
    l.device = &ds;
    note_lease(l);
    return l;
  }
  device_lease lease_any() {
    device_lease l;
    for (auto& d : devices) {
      std::unique_lock<std::mutex> lock(d->mu, std::try_to_lock);
      if (lock.owns_lock()) {
        l.held.push_back(std::move(lock));
        l.device = d.get();
        note_lease(l);
        return l;
      }
    }
    device_state& ds = *devices[next_device.fetch_add(1) % devices.size()];
    l.held.emplace_back(ds.mu);
    l.device = &ds;
    note_lease(l);
    return l;
  }
  device_lease lease_all() {
    device_lease l;
    for (auto& d : devices) l.held.emplace_back(d->mu);
    l.device = devices.empty() ? nullptr : devices[0].get();
    note_lease(l);
    return l;
  }
  size_t host_shards = 1; // SXT_CPU_BACKEND: host threads a call is split over (FORCE_SHARDS)
  bool generator_cache = false; // BLITZAR_AMD_GENERATOR_CACHE
  std::atomic<u64> cache_hits{0}, cache_builds{0};
  device_arena gather; // on devices[0]: partial results of the other devices (row-split calls)
  // RCCL communicators of bzamd_msm_multi_device (one per device slot, ncclCommInitAll on first
  // use); `exchange_state`: 0 = not tried, 1 = RCCL, 2 = peer copies (logical devices sharing a
  // physical one -- RCCL refuses duplicate devices -- or no librccl)
  std::vector<void*> comms;
  int exchange_state = 0;
  void (*destroy_comm)(void*) = nullptr;

  // built-in ristretto generators 0 .. num_precomputed-1
  std::vector<ed_point> host_generators;  // raw extended coordinates
  std::vector<ed_point> host_one_commits; // [i] = g_0 + ... + g_{i-1}

  // engine contexts of devices touched only through the device entry points (bzamd_*_device)
  std::mutex context_mutex;
  std::map<int, msm_context*> device_contexts;

  device_state& primary() { return *devices[0]; }

  // the context asynchronous device entry points use: the one of the CURRENT device
  msm_context* context_for_current_device() {
    int dev = 0;
    BZ_HIP_CHECK(hipGetDevice(&dev));
    for (auto& d : devices) {
      if (d->device == dev) return d->ctx;
    }
    std::lock_guard<std::mutex> lock(context_mutex);
    auto it = device_contexts.find(dev);
    if (it != device_contexts.end()) return it->second;
    msm_context* c = msm_context_new();
    device_contexts.emplace(dev, c);
    return c;
  }

  ~api_state() {
    if (backend == 2) {
      int current = 0;
      (void)hipGetDevice(&current);
      if (!devices.empty()) {
        (void)hipSetDevice(devices[0]->device);
        gather.release();
      }
      if (destroy_comm != nullptr) {
        for (void* c : comms) destroy_comm(c);
      }
      devices.clear();
      for (auto& kv : device_contexts) {
        (void)hipSetDevice(kv.first);
        (void)hipDeviceSynchronize();
        msm_context_free(kv.second);
      }
      (void)hipSetDevice(current);
    }
  }
};

struct resident_generators {
  const curve_vtable* vt = nullptr;
  u64 n = 0;
  resident_table table;
};

api_state* current_state();
} // namespace bz
