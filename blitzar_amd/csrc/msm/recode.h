// Signed radix-2^c recoding of one scalar, shared by the device kernel (k_recode) and the host
// backend.  A scalar is a little-endian bit field of 1..256 bits inside its row (byte-aligned
// `nbytes`-wide for the Pedersen API, arbitrary offsets for the packed fixed-base API,
// cbindings/blitzar_api.h:688-712), two's complement when the column is signed (reference semantics: sxt/multiexp/base/exponent_sequence.h:25-42, abs handling
// sxt/base/num/abs.h:43-53).  Digits D_w satisfy  x = sum_w D_w 2^(c w),  |D_w| <= 2^(c-1),
// and the top digit never carries as long as W * c >= bit_width + 1.
#pragma once

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

// (Every array below is indexed with compile-time constants only: a run-time index into a small
// register array makes hipcc park it in LDS or scratch -- the first version of this walker kept a
// bit cursor into w[] and cost k_recode a dozen LDS round trips per digit.)
struct digit_recoder {
  u64 w[4]; // what is left of |x|: the next digit is its low c bits
  u32 c;
  u32 half;
  u32 carry;
  bool negative;

  // keep bits [0, bit_width) of the 256-bit value t[0..3]
  BZ_HD void keep_field(const u64 t[4], u32 bit_width) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32 lo = 64 * i;
      if (bit_width <= lo) {
        w[i] = 0;
      } else if (bit_width < lo + 64) {
        w[i] = t[i] & ((u64{1} << (bit_width - lo)) - 1);
      } else {
        w[i] = t[i];
      }
    }
  }

  // little-endian bit field [bit_offset, bit_offset + bit_width) of the row at `p`
  BZ_HD void load(const u8* __restrict__ p, u32 bit_offset, u32 bit_width) {
    p += bit_offset >> 3;
    const u32 sh = bit_offset & 7;
    if (sh == 0 && bit_width == 256 && (reinterpret_cast<uintptr_t>(p) & 7) == 0) {
      const u64* q = reinterpret_cast<const u64*>(p);
      w[0] = q[0];
      w[1] = q[1];
      w[2] = q[2];
      w[3] = q[3];
      return;
    }
    const u32 nbytes = (sh + bit_width + 7) >> 3; // <= 33
    u64 t[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (u32 i = 0; i < 33; ++i) {
      if (i < nbytes) t[i >> 3] |= static_cast<u64>(p[i]) << (8 * (i & 7));
    }
    if (sh != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = (t[i] >> sh) | (t[i + 1] << (64 - sh));
    }
    keep_field(t, bit_width);
  }

  // the same field given as ten aligned 32-bit words and the bit position (< 40) of the field's
  // first bit inside them (k_recode_packed reads its LDS tile this way: no byte loads)
  BZ_HD void load_words32(const u32* __restrict__ d, u32 bit_shift, u32 bit_width) {
    u64 t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = d[2 * i] | (static_cast<u64>(d[2 * i + 1]) << 32);
    if (bit_shift != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = (t[i] >> bit_shift) | (t[i + 1] << (64 - bit_shift));
    }
    keep_field(t, bit_width);
  }

  BZ_HD void init(const u8* __restrict__ p, u32 bit_offset, u32 bit_width, bool is_signed,
                  u32 window_bits) {
    load(p, bit_offset, bit_width);
    start(bit_width, is_signed, window_bits);
  }

  BZ_HD void init_words32(const u32* __restrict__ d, u32 bit_shift, u32 bit_width, bool is_signed,
                          u32 window_bits) {
    load_words32(d, bit_shift, bit_width);
    start(bit_width, is_signed, window_bits);
  }

  // w holds the field: set up the digit walk (|x| for negative values of a signed column)
  BZ_HD void start(u32 bit_width, bool is_signed, u32 window_bits) {
    c = window_bits;
    half = 1u << (c - 1);
    carry = 0;
    negative = false;
    if (is_signed) {
      const u32 nbits = bit_width;
      const u32 top = nbits - 1;
      const u32 word = top >> 6;
      const u64 top_word = word == 0 ? w[0] : (word == 1 ? w[1] : (word == 2 ? w[2] : w[3]));
      negative = ((top_word >> (top & 63)) & 1) != 0;
      if (negative) {
        // |x| = 2^nbits - x, computed over the nbits-wide field
        u64 t[4];
        u64 cin = 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u64 v = ~w[i] + cin;
          cin = (cin != 0 && v == 0) ? 1 : 0;
          t[i] = v;
        }
        keep_field(t, nbits);
      }
    }
  }

  // next signed digit, least-significant window first, column sign already applied.  The value
  // is shifted down by c afterwards (c in 1..16, so 64 - c is a valid shift count).
  BZ_HD int next() {
    const u32 u = static_cast<u32>(w[0]) & ((1u << c) - 1);
    w[0] = (w[0] >> c) | (w[1] << (64 - c));
    w[1] = (w[1] >> c) | (w[2] << (64 - c));
    w[2] = (w[2] >> c) | (w[3] << (64 - c));
    w[3] >>= c;
    const u32 t = u + carry;
    int d;
    if (t > half) {
      d = static_cast<int>(t) - static_cast<int>(1u << c);
      carry = 1;
    } else {
      d = static_cast<int>(t);
      carry = 0;
    }
    return negative ? -d : d;
  }
};
} // namespace bz
