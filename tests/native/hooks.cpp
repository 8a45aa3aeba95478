// Test-only C hooks over the product's header-only field/curve code (host compilation), so that
// the CPU test-suite can compare it limb-for-limb with the reference oracle (oracle/_ref).
// Built by __graft_entry__.build() into tests/native/_build/libbz_hooks.so.
#define BZ_MONT29_CHECK 1
#include <cstring>

#include "blitzar_amd/csrc/curve/ed25519.h"
#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/curve/sw29.h"
#include "blitzar_amd/csrc/curve/weierstrass.h"
#include "blitzar_amd/csrc/msm/plan.h"
#include "blitzar_amd/csrc/msm/recode.h"

using namespace bz;

extern "C" {
void bz_f51_mul(u64* h, const u64* f, const u64* g) {
  fe51 a, b;
  std::memcpy(&a, f, 40);
  std::memcpy(&b, g, 40);
  fe51 r = f51::mul(a, b);
  std::memcpy(h, &r, 40);
}
void bz_f51_sq(u64* h, const u64* f) {
  fe51 a;
  std::memcpy(&a, f, 40);
  fe51 r = f51::sq(a);
  std::memcpy(h, &r, 40);
}
void bz_f51_sub(u64* h, const u64* f, const u64* g) {
  fe51 a, b;
  std::memcpy(&a, f, 40);
  std::memcpy(&b, g, 40);
  fe51 r = f51::sub(a, b);
  std::memcpy(h, &r, 40);
}
void bz_f51_invert(u64* h, const u64* f) {
  fe51 a;
  std::memcpy(&a, f, 40);
  fe51 r = f51::invert(a);
  std::memcpy(h, &r, 40);
}
void bz_ed_base_elements(u64* out, u64 first, u64 n) {
  for (u64 i = 0; i < n; ++i) {
    ed_point p = ed::base_element(first + i);
    std::memcpy(out + 20 * i, &p, 160);
  }
}
void bz_ed_add(u64* out, const u64* a, const u64* b) {
  ed_point p, q;
  std::memcpy(&p, a, 160);
  std::memcpy(&q, b, 160);
  ed_point r = ed::add(p, q);
  std::memcpy(out, &r, 160);
}
void bz_ed_dbl(u64* out, const u64* a) {
  ed_point p;
  std::memcpy(&p, a, 160);
  ed_point r = ed::dbl(p);
  std::memcpy(out, &r, 160);
}
void bz_ed_dbl_n(u64* out, const u64* a, int k) {
  ed_point p;
  std::memcpy(&p, a, 160);
  ed_point r = ed::dbl_n(p, k);
  std::memcpy(out, &r, 160);
}
void bz_ed_neg(u64* out, const u64* a) {
  ed_point p;
  std::memcpy(&p, a, 160);
  ed_point r = ed::neg(p);
  std::memcpy(out, &r, 160);
}
void bz_ed_sub_cached(u64* out, const u64* a, const u64* b) {
  ed_point p, q;
  std::memcpy(&p, a, 160);
  std::memcpy(&q, b, 160);
  ed_point r = ed::to_point(ed::sub_cached(p, ed::to_cached(q)));
  std::memcpy(out, &r, 160);
}
void bz_ristretto_encode(u8* out, const u64* a) {
  ed_point p;
  std::memcpy(&p, a, 160);
  ristretto::encode(out, p);
}
int bz_ristretto_decode(u64* out, const u8* s) {
  ed_point p;
  bool ok = ristretto::decode(p, s);
  std::memcpy(out, &p, 160);
  return ok ? 0 : -1;
}

#define BZ_SW_HOOKS(PFX, G)                                                                        \
  void bz_##PFX##_field_mul(u64* h, const u64* f, const u64* g) {                                  \
    G::fe a, b;                                                                                    \
    std::memcpy(&a, f, sizeof(a));                                                                 \
    std::memcpy(&b, g, sizeof(b));                                                                 \
    G::fe r = G::F::mul(a, b);                                                                     \
    std::memcpy(h, &r, sizeof(r));                                                                 \
  }                                                                                                \
  void bz_##PFX##_field_addsub(u64* s, u64* d, const u64* f, const u64* g) {                       \
    G::fe a, b;                                                                                    \
    std::memcpy(&a, f, sizeof(a));                                                                 \
    std::memcpy(&b, g, sizeof(b));                                                                 \
    G::fe r = G::F::add(a, b);                                                                     \
    std::memcpy(s, &r, sizeof(r));                                                                 \
    r = G::F::sub(a, b);                                                                           \
    std::memcpy(d, &r, sizeof(r));                                                                 \
  }                                                                                                \
  void bz_##PFX##_add(u64* out, const u64* a, const u64* b) {                                      \
    G::point p, q;                                                                                 \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    std::memcpy(&q, b, sizeof(q));                                                                 \
    G::point r = G::add(p, q);                                                                     \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  void bz_##PFX##_add_mixed(u64* out, const u64* a, const u64* b_affine) {                         \
    G::point p;                                                                                    \
    G::affine q;                                                                                   \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    std::memcpy(&q, b_affine, sizeof(q));                                                          \
    G::point r = G::add_mixed(p, q);                                                               \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  void bz_##PFX##_dbl(u64* out, const u64* a) {                                                    \
    G::point p;                                                                                    \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    G::point r = G::dbl(p);                                                                        \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  int bz_##PFX##_to_affine(u64* out_xy, const u64* a) {                                            \
    G::point p;                                                                                    \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    G::affine r;                                                                                   \
    bool inf = G::to_affine(r, p);                                                                 \
    std::memcpy(out_xy, &r, sizeof(r));                                                            \
    return inf ? 1 : 0;                                                                            \
  }
BZ_SW_HOOKS(bn254, bn254_g1)
BZ_SW_HOOKS(grumpkin, grumpkin_g)
BZ_SW_HOOKS(bls12_381, bls12_381_g1)

void bz_bls12_381_compress(u8* out48, const u64* a) {
  bls12_381_g1::point p;
  std::memcpy(&p, a, sizeof(p));
  bls12_381_g1_compress(out48, p);
}

// signed radix-2^c digits of one scalar bit field (msm/recode.h); returns the number of digits
int bz_recode(int* digits, const u8* row, u32 bit_offset, u32 bit_width, int is_signed,
              u32 window_bits, u32 num_windows) {
  digit_recoder rec;
  rec.init(row, bit_offset, bit_width, is_signed != 0, window_bits);
  for (u32 w = 0; w < num_windows; ++w) digits[w] = rec.next();
  return static_cast<int>(num_windows);
}

// digit_recoder::load_words32 (the LDS-tile path of k_recode_packed) against ::load: the field of
// `bit_width` bits at `bit_offset` of the row, read from ten aligned words at byte `skew` (0..3)
int bz_recode_words32(int* digits, const u8* row, u32 skew, u32 bit_offset, u32 bit_width,
                      int is_signed, u32 window_bits, u32 num_windows) {
  u32 words[12] = {};
  // the field starts at byte (bit_offset >> 3) of `row`; place that byte at `skew` in the words
  std::memcpy(reinterpret_cast<u8*>(words) + skew, row + (bit_offset >> 3), 34);
  digit_recoder rec;
  rec.init_words32(words, 8 * skew + (bit_offset & 7), bit_width, is_signed != 0, window_bits);
  for (u32 w = 0; w < num_windows; ++w) digits[w] = rec.next();
  return static_cast<int>(num_windows);
}

// planner (msm/plan.h): per column {window_bits, num_windows, slices, first_task, slice_rows,
// group_bits}; totals {tasks, total_buckets, total_entries, total_segments, rows covered,
// total_groups, log2 entries per accumulate lane, log2 buckets per reduce lane}
void bz_plan(u32* per_column, u64* totals, const u64* n, const u32* bit_width, const int* is_signed,
             u32 num_columns, u32 max_window_bits, int in_sequence) {
  std::vector<host_column> cols(num_columns);
  for (u32 i = 0; i < num_columns; ++i) {
    cols[i] = host_column{nullptr, n[i], (bit_width[i] + 7) / 8, 0, bit_width[i], is_signed[i] != 0};
  }
  msm_tuning tune;
  tune.max_window_bits = max_window_bits;
  tune.in_sequence = in_sequence != 0; // the call runs in throughput mode (msm_enqueue sets this)
  msm_plan plan = make_msm_plan(cols, tune);
  for (u32 i = 0; i < num_columns; ++i) {
    const column_desc& c = plan.columns[i];
    per_column[6 * i + 0] = c.window_bits;
    per_column[6 * i + 1] = c.num_windows;
    per_column[6 * i + 2] = c.num_windows == 0 ? 0 : plan.tasks[c.first_task].num_slices;
    per_column[6 * i + 3] = c.first_task;
    per_column[6 * i + 4] = c.num_windows == 0 ? 0 : plan.tasks[c.first_task].slice_rows;
    per_column[6 * i + 5] = c.num_windows == 0 ? 0 : plan.tasks[c.first_task].group_bits;
  }
  totals[0] = plan.tasks.size();
  totals[1] = plan.total_buckets;
  totals[2] = plan.total_entries;
  totals[3] = plan.total_segments;
  u64 covered = 0;
  for (const auto& t : plan.tasks) covered += t.rows;
  totals[4] = covered;
  totals[5] = plan.total_groups;
  totals[6] = plan.segment_log2;
  totals[7] = plan.reduce_segment_log2;
}

// planner with a window table (msm/plan.h `window_table`): per column {window_bits, num_windows,
// num_tasks, merged_stride, rows of the column's first task (low / high 32 bits)}; totals {tasks,
// total_buckets, total_entries, max_task_rows, max_recode_rows, max_rows, wide_digits, group bits and
// groups of the first merged task}
void bz_plan_tables(u32* per_column, u64* totals, const u64* n, const u32* bit_width,
                    const int* is_signed, u32 num_columns, u64 stride, u32 windows, int force,
                    double table_penalty, u32 table_bits) {
  std::vector<host_column> cols(num_columns);
  for (u32 i = 0; i < num_columns; ++i) {
    cols[i] = host_column{nullptr, n[i], (bit_width[i] + 7) / 8, 0, bit_width[i], is_signed[i] != 0};
  }
  msm_tuning tune;
  tune.force_window_tables = force != 0;
  tune.table_penalty = table_penalty;
  window_table tables;
  tables.stride = stride;
  tables.windows = windows;
  tables.bits = table_bits;
  msm_plan plan = make_msm_plan(cols, tune, &tables);
  for (u32 i = 0; i < num_columns; ++i) {
    const column_desc& c = plan.columns[i];
    const u64 rows = c.num_tasks == 0 ? 0 : plan.tasks[c.first_task].rows;
    per_column[6 * i + 0] = c.window_bits;
    per_column[6 * i + 1] = c.num_windows;
    per_column[6 * i + 2] = c.num_tasks;
    per_column[6 * i + 3] = c.merged_stride;
    per_column[6 * i + 4] = static_cast<u32>(rows);
    per_column[6 * i + 5] = static_cast<u32>(rows >> 32);
  }
  totals[0] = plan.tasks.size();
  totals[1] = plan.total_buckets;
  totals[2] = plan.total_entries;
  totals[3] = plan.max_task_rows;
  totals[4] = plan.max_recode_rows;
  totals[5] = plan.max_rows;
  totals[6] = plan.wide_digits ? 1 : 0;
  totals[7] = totals[8] = 0;
  for (u32 i = 0; i < num_columns; ++i) {
    const column_desc& c = plan.columns[i];
    if (c.merged_stride != 0) {
      totals[7] = plan.tasks[c.first_task].group_bits;
      totals[8] = plan.tasks[c.first_task].num_groups;
      break;
    }
  }
}

// choose_call_table (msm/plan.h): out = {stride, windows, bits}, costs = {separate, merged, build}
void bz_choose_call_table(u64* out, double* costs, const u64* n, const u32* bit_width,
                          const int* is_signed, u32 num_columns, u64 addend_size, double entry_cost,
                          u32 force_bits) {
  std::vector<host_column> cols(num_columns);
  for (u32 i = 0; i < num_columns; ++i) {
    cols[i] = host_column{nullptr, n[i], (bit_width[i] + 7) / 8, 0, bit_width[i], is_signed[i] != 0};
  }
  const call_table_choice ch = choose_call_table(cols, msm_tuning{}, addend_size, entry_cost, force_bits);
  out[0] = ch.shape.stride;
  out[1] = ch.shape.windows;
  out[2] = ch.shape.bits;
  costs[0] = ch.separate_cost;
  costs[1] = ch.merged_cost;
  costs[2] = ch.build_cost;
}

// column ranges of k_recode_packed (msm/plan.h): columns = bit fields at byte offsets `offset[i]`
// of rows `stride[i]` bytes apart; out = {first_column, num_columns, base offset, span} per range;
// returns the number of ranges (0: not a packed batch)
int bz_packed_ranges(u32* out, const u64* offset, const u64* stride, const u32* bit_offset,
                     const u32* bit_width, u32 num_columns) {
  static u8 arena[1];
  msm_plan plan;
  for (u32 i = 0; i < num_columns; ++i) {
    column_desc c{};
    c.data = offset[i] == ~u64{0} ? nullptr : arena + offset[i];
    c.n = 100;
    c.row_stride = stride[i];
    c.bit_offset = bit_offset[i];
    c.bit_width = bit_width[i];
    plan.columns.push_back(c);
  }
  const auto ranges = packed_recode_ranges(plan);
  for (size_t k = 0; k < ranges.size(); ++k) {
    out[4 * k + 0] = ranges[k].first_column;
    out[4 * k + 1] = ranges[k].num_columns;
    out[4 * k + 2] = static_cast<u32>(ranges[k].base - arena);
    out[4 * k + 3] = ranges[k].span;
  }
  return static_cast<int>(ranges.size());
}

// 9 x 29-bit field of the gfx950 kernels (field/f29.h); limbs in/out are raw u32[9]
void bz_f29_from_fe51(u32* h, const u64* f) {
  fe51 a;
  std::memcpy(&a, f, 40);
  fe29 r = f29::from_fe51(a);
  std::memcpy(h, &r, 36);
}
void bz_f29_to_words(u64* w, const u32* f) {
  fe29 a;
  std::memcpy(&a, f, 36);
  f29::to_words(w, a);
}
void bz_f29_mul(u32* h, const u32* f, const u32* g) {
  fe29 a, b;
  std::memcpy(&a, f, 36);
  std::memcpy(&b, g, 36);
  fe29 r = f29::mul(a, b);
  std::memcpy(h, &r, 36);
}
void bz_f29_sq(u32* h, const u32* f) {
  fe29 a;
  std::memcpy(&a, f, 36);
  fe29 r = f29::sq(a);
  std::memcpy(h, &r, 36);
}
void bz_f29_sub(u32* h, const u32* f, const u32* g) {
  fe29 a, b;
  std::memcpy(&a, f, 36);
  std::memcpy(&b, g, 36);
  fe29 r = f29::sub(a, b);
  std::memcpy(h, &r, 36);
}
void bz_f29_weak_reduce(u32* h, const u32* f) {
  fe29 a;
  std::memcpy(&a, f, 36);
  fe29 r = f29::weak_reduce(a);
  std::memcpy(h, &r, 36);
}
void bz_f29_invert(u32* h, const u32* f) {
  fe29 a;
  std::memcpy(&a, f, 36);
  fe29 r = f29::invert(a);
  std::memcpy(h, &r, 36);
}
// out = a + b (or a - b) computed on the 29-bit representation, returned as fe51 element_p3
void bz_ed29_add(u64* out, const u64* a, const u64* b, int negate) {
  ed_point p, q;
  std::memcpy(&p, a, 160);
  std::memcpy(&q, b, 160);
  ed29_point r = ed29::add_cached(ed29::from_ed(p), ed29::cached_from_ed(q), negate != 0);
  ed_point e = ed29::to_ed(r);
  std::memcpy(out, &e, 160);
}
// the same through what k_accumulate does with a packed row: gathered with its first two pieces
// exchanged by address when negating, unpacked, 2dT negated by xor-add
void bz_ed29_add_gathered(u64* out, const u64* a, const u64* b, int negate) {
  ed_point p, q;
  std::memcpy(&p, a, 160);
  std::memcpy(&q, b, 160);
  const ed29_cached_packed row[1] = {ed29::pack(ed29::cached_from_ed(q))};
  const ed29_cached_packed g = ed29::gather_signed(row, 0, negate != 0);
  const ed29_point acc =
      ed29::add_cached_presigned(ed29::from_ed(p), ed29::unpack(g), negate != 0);
  ed_point e = ed29::to_ed(acc);
  std::memcpy(out, &e, 160);
}
void bz_ed29_dbl_n(u64* out, const u64* a, int k) {
  ed_point p;
  std::memcpy(&p, a, 160);
  ed_point e = ed29::to_ed(ed29::dbl_n(ed29::from_ed(p), k));
  std::memcpy(out, &e, 160);
}
// the same chain through the Z = 1 resident addends (ed29_niels, 7-product addition)
void bz_ed29_chain_niels(u64* out, const u64* points, const int* negate, int n) {
  ed29_point acc = ed29::identity();
  for (int i = 0; i < n; ++i) {
    ed_point q;
    std::memcpy(&q, points + 20 * i, 160);
    acc = ed29::add_niels(acc, ed29::to_niels(ed29::from_ed(q)), negate[i] != 0);
  }
  ed_point e = ed29::to_ed(acc);
  std::memcpy(out, &e, 160);
}
// long chain: acc = sum_i (+-) q_i, stressing the limb bounds of repeated accumulation
void bz_ed29_chain(u64* out, const u64* points, const int* negate, int n) {
  ed29_point acc = ed29::identity();
  for (int i = 0; i < n; ++i) {
    ed_point q;
    std::memcpy(&q, points + 20 * i, 160);
    acc = ed29::add_cached(acc, ed29::cached_from_ed(q), negate[i] != 0);
  }
  ed_point e = ed29::to_ed(acc);
  std::memcpy(out, &e, 160);
}

// what a lane of k_accumulate does with its segment: the first entry loaded (from_cached_presigned /
// from_niels), the others added
void bz_ed29_chain_first(u64* out, const u64* points, const int* negate, int n, int niels) {
  ed29_point acc = ed29::identity();
  for (int i = 0; i < n; ++i) {
    ed_point q;
    std::memcpy(&q, points + 20 * i, 160);
    const bool neg = negate[i] != 0;
    if (niels != 0) {
      const ed29_niels row = ed29::to_niels(ed29::from_ed(q));
      acc = i == 0 ? ed29::from_niels(row, neg) : ed29::add_niels(acc, row, neg);
    } else {
      const ed29_cached_packed row[1] = {ed29::pack(ed29::cached_from_ed(q))};
      const ed29_cached g = ed29::unpack(ed29::gather_signed(row, 0, neg));
      acc = i == 0 ? ed29::from_cached_presigned(g, neg) : ed29::add_cached_presigned(acc, g, neg);
    }
  }
  ed_point e = ed29::to_ed(acc);
  std::memcpy(out, &e, 160);
}

// unsaturated-limb Montgomery fields / curves of the gfx950 kernels (field/mont29.h,
// curve/sw29.h), driven through their ABI-form conversions; built with BZ_MONT29_CHECK so every
// limb-level contract is asserted while the tests run
#define BZ_SW29_HOOKS(PFX, G)                                                                      \
  void bz_##PFX##_29_field_mul(u64* h, const u64* f, const u64* g) {                               \
    const auto a = G::F::from_mont64(f), b = G::F::from_mont64(g);                                  \
    const auto slow = G::F::mul(a, b), fast = G::F::mul_pinned(a, b);                              \
    if (std::memcmp(&slow, &fast, sizeof(slow)) != 0) std::abort(); /* product scanning == */      \
    G::F::to_mont64(h, slow);                                                                      \
  }                                                                                                \
  void bz_##PFX##_29_field_roundtrip(u64* h, const u64* f) {                                       \
    G::F::to_mont64(h, G::F::from_mont64(f));                                                      \
  }                                                                                                \
  void bz_##PFX##_29_field_invert(u64* h, const u64* f) {                                          \
    G::F::to_mont64(h, G::F::invert(G::F::from_mont64(f)));                                        \
  }                                                                                                \
  /* (2a) b + c (3d) by one fused reduction (mul2): lazily added operands, B 2 x 1 + 1 x 3 */       \
  void bz_##PFX##_29_field_mul2(u64* h, const u64* a, const u64* b, const u64* c, const u64* d) {  \
    using F = G::F;                                                                                \
    const auto fa = F::from_mont64(a), fb = F::from_mont64(b), fc = F::from_mont64(c),             \
               fd = F::from_mont64(d);                                                             \
    const auto lhs = F::mul2(F::add(fa, fa), fb, fc, F::add(F::add(fd, fd), fd));                  \
    const auto scan = F::mul2_pinned(F::add(fa, fa), fb, fc, F::add(F::add(fd, fd), fd));          \
    if (std::memcmp(&lhs, &scan, sizeof(lhs)) != 0) std::abort();                                  \
    F::to_mont64(h, lhs);                                                                          \
  }                                                                                                \
  void bz_##PFX##_29_add(u64* out, const u64* a, const u64* b) {                                   \
    G::G64::point p, q;                                                                            \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    std::memcpy(&q, b, sizeof(q));                                                                 \
    G::G64::point r = G::to_point64(G::add(G::from_point64(p), G::from_point64(q)));               \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  void bz_##PFX##_29_dbl_n(u64* out, const u64* a, int k) {                                        \
    G::G64::point p;                                                                               \
    std::memcpy(&p, a, sizeof(p));                                                                 \
    G::G64::point r = G::to_point64(G::dbl_n(G::from_point64(p), k));                              \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  /* acc = start + sum_i (+-) q_i with mixed additions; q_i affine {x, y} ABI form */              \
  void bz_##PFX##_29_chain(u64* out, const u64* start, const u64* affine_xy, const int* negate,    \
                           int n) {                                                                \
    G::G64::point p;                                                                               \
    std::memcpy(&p, start, sizeof(p));                                                             \
    G::point acc = G::from_point64(p);                                                             \
    /* k_accumulate's form (add_mixed_acc, tighter invariant, fewer partial reductions): the same  \
       residues coordinate by coordinate; its invariant is asserted inside (check build) */         \
    G::point tight = acc;                                                                          \
    constexpr int W = G::N64;                                                                      \
    for (int i = 0; i < n; ++i) {                                                                  \
      G::affine q = G::affine_from_mont64(affine_xy + 2 * W * i, affine_xy + 2 * W * i + W, false); \
      /* alternate between the two forms of the field products: both must give the same limbs */    \
      const G::point fast = G::template add_mixed<true>(acc, q, negate[i] != 0);                   \
      acc = G::add_mixed(acc, q, negate[i] != 0);                                                  \
      if (std::memcmp(&fast, &acc, sizeof(acc)) != 0) std::abort();                                \
      const G::point tight_slow = G::template add_mixed_acc<false>(tight, q, negate[i] != 0);      \
      tight = G::template add_mixed_acc<true>(tight, q, negate[i] != 0);                           \
      if (std::memcmp(&tight_slow, &tight, sizeof(tight)) != 0) std::abort();                      \
    }                                                                                              \
    G::G64::point r = G::to_point64(acc);                                                          \
    const G::G64::point rt = G::to_point64(tight);                                                 \
    if (std::memcmp(&r, &rt, sizeof(r)) != 0) std::abort();                                        \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }
/* a lane of k_accumulate: the first affine entry lifted (lift_acc), the others added */
#define BZ_SW29_LIFT_HOOK(PFX, G)                                                                  \
  void bz_##PFX##_29_chain_lifted(u64* out, const u64* affine_xy, const int* negate, int n) {      \
    constexpr int W = G::N64;                                                                      \
    G::point acc = G::identity();                                                                  \
    for (int i = 0; i < n; ++i) {                                                                  \
      G::affine q = G::affine_from_mont64(affine_xy + 2 * W * i, affine_xy + 2 * W * i + W, false); \
      acc = i == 0 ? G::lift_acc(q, negate[i] != 0)                                                \
                   : G::template add_mixed_acc<true>(acc, q, negate[i] != 0);                      \
    }                                                                                              \
    G::G64::point r = G::to_point64(acc);                                                          \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }
BZ_SW29_LIFT_HOOK(bn254, bn254_g1_29)
BZ_SW29_LIFT_HOOK(grumpkin, grumpkin_29)
BZ_SW29_LIFT_HOOK(bls12_381, bls12_381_g1_28)
BZ_SW29_HOOKS(bn254, bn254_g1_29)
BZ_SW29_HOOKS(grumpkin, grumpkin_29)
BZ_SW29_HOOKS(bls12_381, bls12_381_g1_28)
}
