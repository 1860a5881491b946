// Host build of the per-segment DEFLATE decode that every GPU lane runs (starflate_amd/csrc/sf_inflate_core.h),
// so its logic is testable in the CPU suite.  TEST INFRASTRUCTURE: built into a throw-away .so by
// tests/test_inflate_core_host.py; the product never loads it (the GPU kernels are the only decoder shipped).
#include "../../starflate_amd/csrc/sf_inflate_core.h"

#include <cstdlib>
#include <cstring>

extern "C" {

// -> status; tokens_out: room for 32768 tokens (16-byte aligned by the caller)
unsigned sfi_decode_segment(const unsigned char* src, unsigned long long src_n, unsigned long long seg_begin,
                            unsigned long long seg_end, unsigned out_n, unsigned* tokens_out, unsigned* ntok,
                            unsigned* raw, unsigned long long* raw_off) {
  alignas(16) static thread_local unsigned char mem[sf::inflate::LaneLayout::kBytes + 12];
  std::memset(mem, 0xA5, sizeof mem);  // stale table contents must not matter
  const auto r = sf::inflate::decode_segment(src, src_n, seg_begin, seg_end, out_n, tokens_out, mem);
  *ntok = r.ntok;
  *raw = r.raw;
  *raw_off = r.raw_off;
  return r.status;
}

// the sub-indexed path of one segment, region after region as the 32 GPU lanes would: tokens land at their
// compact positions (sub[2r+1] + k).  -> status; *ntok = tokens of the segment
unsigned sfi_decode_segment_sub(const unsigned char* src, unsigned long long src_n, unsigned long long seg_begin,
                                unsigned long long seg_end, unsigned out_n, const unsigned* sub /*[32][2]*/,
                                unsigned* tokens_out, unsigned* ntok, unsigned* raw, unsigned long long* raw_off) {
  alignas(16) static thread_local unsigned char mem[sf::inflate::LaneLayout::kBytes + 12];
  std::memset(mem, 0x5A, sizeof mem);
  uint64_t hdr_end = 0, roff = 0;
  uint32_t is_raw = 0;
  unsigned st = sf::inflate::open_segment<sf::inflate::LaneLayout>(src, src_n, seg_begin, seg_end, out_n, mem, true, is_raw, roff, hdr_end);
  *raw = is_raw;
  *raw_off = roff;
  *ntok = 0;
  if (st || *raw) return st;
  if (sub[0] != hdr_end) return sf::inflate::kError;
  for (unsigned r = 0; r < 32 && !st; ++r) {
    const unsigned ob = r * 1024 < out_n ? r * 1024 : out_n, oe = (r + 1) * 1024 < out_n ? (r + 1) * 1024 : out_n;
    if (sub[2 * r + 1] > ob) return sf::inflate::kError;
    uint32_t n = 0;
    st = sf::inflate::decode_region<sf::inflate::LaneLayout>(src, src_n, seg_begin, seg_end, sub[2 * r], r < 31 ? sub[2 * r + 2] : 0, r == 31, ob, oe,
                                    tokens_out + sub[2 * r + 1], mem, n);
    if (!st && r < 31 && sub[2 * r + 1] + n != sub[2 * r + 3]) st = sf::inflate::kError;
    if (r == 31) *ntok = sub[2 * r + 1] + n;
  }
  return st;
}

// scalar token expansion (what k_inflate_bytes does with a wave): -> bytes written, or -1 on a bad token
long long sfi_expand_tokens(const unsigned* tokens, unsigned ntok, unsigned char* out, unsigned out_n) {
  unsigned pos = 0;
  for (unsigned i = 0; i < ntok; ++i) {
    const unsigned t = tokens[i];
    if (t & 0x80000000u) {
      const unsigned len = ((t >> 16) & 0xFF) + 3, dist = (t & 0x7FFF) + 1;
      if (dist > pos || len > out_n - pos) return -1;
      for (unsigned k = 0; k < len; ++k, ++pos) out[pos] = out[pos - dist];
    } else {
      if (pos >= out_n) return -1;
      out[pos++] = (unsigned char)t;
    }
  }
  return pos;
}
}
