// Merlin transcripts over STROBE-128 / Keccak-f[1600], operating in place on the caller's
// 203-byte `sxt_transcript` (cbindings/blitzar_api.h:61-63 = prft::transcript: the 200-byte
// sponge state followed by pos, pos_begin, cur_flags; sxt/proof/transcript/strobe128.h:41-45).
// Behaviour restated from the reference (sxt/proof/transcript/strobe128.cc, transcript.cc,
// transcript_utility.h, themselves ports of libmerlin): the byte-for-byte state after every
// operation is observable by the caller, who continues the same transcript after the proof.
// Host only.
#pragma once

#include <cstdint>
#include <cstring>
#include <string_view>

#include "blitzar_amd/csrc/base/macros.h"

namespace bz::proof {

// Keccak-f[1600] on 25 little-endian 64-bit lanes (FIPS 202, section 3.3), lane (x, y) at a[x + 5 y]
inline void keccak_f1600(u64 a[25]) {
  static constexpr u64 round_constant[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
      0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
      0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  // rotation offsets r[x][y] (FIPS 202 table 2)
  static constexpr unsigned rotation[5][5] = {{0, 36, 3, 41, 18},
                                              {1, 44, 10, 45, 2},
                                              {62, 6, 43, 15, 61},
                                              {28, 55, 25, 21, 56},
                                              {27, 20, 39, 8, 14}};
  auto rotl = [](u64 v, unsigned s) { return s == 0 ? v : (v << s) | (v >> (64 - s)); };
  for (int round = 0; round < 24; ++round) {
    // theta
    u64 parity[5];
    for (int x = 0; x < 5; ++x) {
      parity[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    }
    for (int x = 0; x < 5; ++x) {
      const u64 d = parity[(x + 4) % 5] ^ rotl(parity[(x + 1) % 5], 1);
      for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
    }
    // rho and pi: b[y][2x + 3y] = rot(a[x][y], r[x][y])
    u64 b[25];
    for (int x = 0; x < 5; ++x) {
      for (int y = 0; y < 5; ++y) {
        b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], rotation[x][y]);
      }
    }
    // chi
    for (int y = 0; y < 5; ++y) {
      for (int x = 0; x < 5; ++x) {
        a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
      }
    }
    // iota
    a[0] ^= round_constant[round];
  }
}

// the caller's transcript, viewed in place
struct transcript_state {
  u8 state[200];
  u8 pos;
  u8 pos_begin;
  u8 cur_flags;
};
static_assert(sizeof(transcript_state) == 203);

class strobe128 {
public:
  static constexpr u8 kRate = 166; // STROBE-128: 200 - 2 * 128 / 8 - 2
  static constexpr u8 kFlagI = 1, kFlagA = 2, kFlagC = 4, kFlagT = 8, kFlagM = 16, kFlagK = 32;

  explicit strobe128(transcript_state* s) : s_{s} {}

  // a fresh STROBE-128 state with the given protocol label (strobe128.cc:70-73; the initial block
  // is [1, R + 2, 1, 0, 1, 96] || "STROBEv1.0.2", strobe128.h:41-42)
  static void init(transcript_state* s, std::string_view label) {
    std::memset(s, 0, sizeof(*s));
    static constexpr u8 header[18] = {1,  168, 1,  0,  1,   96, 83, 84, 82,
                                      79, 66,  69, 118, 49, 46, 48, 46, 50};
    std::memcpy(s->state, header, sizeof(header));
    permute(s);
    strobe128 st{s};
    st.meta_ad(reinterpret_cast<const u8*>(label.data()), label.size(), false);
  }

  void meta_ad(const u8* data, size_t n, bool more) {
    begin_op(kFlagM | kFlagA, more);
    absorb(data, n);
  }
  void ad(const u8* data, size_t n, bool more) {
    begin_op(kFlagA, more);
    absorb(data, n);
  }
  void prf(u8* out, size_t n, bool more) {
    begin_op(kFlagI | kFlagA | kFlagC, more);
    squeeze(out, n);
  }

private:
  transcript_state* s_;

  static void permute(transcript_state* s) {
    u64 lanes[25];
    std::memcpy(lanes, s->state, 200); // little-endian host (the ABI is little-endian throughout)
    keccak_f1600(lanes);
    std::memcpy(s->state, lanes, 200);
  }
  // strobe128.cc:112-124
  void run_f() {
    s_->state[s_->pos] ^= s_->pos_begin;
    s_->state[s_->pos + 1] ^= 0x04;
    s_->state[kRate + 1] ^= 0x80;
    permute(s_);
    s_->pos = 0;
    s_->pos_begin = 0;
  }
  void absorb(const u8* data, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      s_->state[s_->pos] ^= data[i];
      if (++s_->pos == kRate) run_f();
    }
  }
  void squeeze(u8* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      out[i] = s_->state[s_->pos];
      s_->state[s_->pos] = 0;
      if (++s_->pos == kRate) run_f();
    }
  }
  // strobe128.cc:142-166
  void begin_op(u8 flags, bool more) {
    if (more) return; // continuing the previous operation
    const u8 old_begin = s_->pos_begin;
    s_->pos_begin = static_cast<u8>(s_->pos + 1);
    s_->cur_flags = flags;
    const u8 frame[2] = {old_begin, flags};
    absorb(frame, 2);
    if ((flags & (kFlagC | kFlagK)) != 0 && s_->pos != 0) run_f();
  }
};

// Merlin (transcript.cc:48-88, transcript_utility.h)
class transcript {
public:
  explicit transcript(void* caller_bytes)
      : s_{static_cast<transcript_state*>(caller_bytes)} {}

  // prft::transcript{label}: what an API consumer constructs before calling the prover
  static void init(void* caller_bytes, std::string_view label) {
    auto* s = static_cast<transcript_state*>(caller_bytes);
    strobe128::init(s, "Merlin v1.0");
    transcript t{caller_bytes};
    t.append_message("dom-sep", reinterpret_cast<const u8*>(label.data()), label.size());
  }

  void append_message(std::string_view label, const u8* message, size_t n) {
    const u32 len = static_cast<u32>(n);
    strobe128 st{s_};
    st.meta_ad(reinterpret_cast<const u8*>(label.data()), label.size(), false);
    st.meta_ad(reinterpret_cast<const u8*>(&len), sizeof(len), true);
    st.ad(message, n, false);
  }
  void challenge_bytes(u8* dest, size_t n, std::string_view label) {
    const u32 len = static_cast<u32>(n);
    strobe128 st{s_};
    st.meta_ad(reinterpret_cast<const u8*>(label.data()), label.size(), false);
    st.meta_ad(reinterpret_cast<const u8*>(&len), sizeof(len), true);
    st.prf(dest, n, false);
  }
  void set_domain(std::string_view name) {
    append_message("domain-sep", reinterpret_cast<const u8*>(name.data()), name.size());
  }
  void append_u64(std::string_view label, u64 v) {
    append_message(label, reinterpret_cast<const u8*>(&v), sizeof(v));
  }

private:
  transcript_state* s_;
};
} // namespace bz::proof
