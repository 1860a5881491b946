// sf_inflate_core.h -- one DEFLATE segment -> tokens, written once for device and host.
//
// This is the bit-serial half of the GPU decoder: the inverse of k_emit, i.e. what
// the reference's decompress() does per block (/root/reference/src/decompress.cpp:402-461: header
// :370-385, stored :416-436, fixed tables :25-40, dynamic tables :253-367, symbol loop :122-187),
// minus the byte copies -- a match becomes a token (bit31, len-3 in 16..23, dist-1 in 0..14), a literal
// its byte value, exactly the token format k_lz77 writes.  On the GPU every LANE runs this function on its
// own segment (k_inflate_tokens, sf_inflate.hip); the table memory `m` is that lane's slice of LDS.
// The same source compiles for the host so that tests can run it without a GPU (tests/cpp/inflate_core_host.cpp);
// the product never calls the host build.
//
// Differences from the reference, all on inputs where the reference is undefined or wrong:
//  * the HLIT+HDIST code lengths are one sequence (RFC 1951 3.2.7: a run may cross from the literal/length
//    into the distance lengths); the reference reads two sequences (hazards A/B of SURVEY.md 8(c)).  Streams
//    made by this library never contain such a run, so both decoders agree on them.
//  * truncated input is SrcTooSmall and a code-length run past the end is Error, where the reference asserts.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__ __forceinline__
#else
#define SF_HD inline
#endif

namespace sf {
namespace inflate {

// DecompressStatus of the reference (src/decompress.hpp:13-23), same values
enum : uint32_t {
  kOk = 0,
  kError = 1,
  kInvalidBlockHeader = 2,
  kNoCompressionLenMismatch = 3,
  kDstTooSmall = 4,
  kSrcTooSmall = 5,
  kInvalidLitOrLen = 6,
  kInvalidDistance = 7
};

constexpr uint32_t kTokMatchBit = 0x80000000u;

// Table memory of one segment (byte offsets).  Both code tables: `fast` u16[1 << bits] indexed by the next
// stream bits, entry (symbol << 4) | code bits, 0 = longer code or none; `sym` u16[] symbols sorted by
// (code bits, symbol) and `cnt` u16[16] codes per length for the bit-serial walk of longer codes; `lens` u8[320]
// code lengths (literal/length at 0, distance at 288).  While a dynamic header is parsed the literal/length
// fast table is not built yet and lends its first 160 bytes to the code-length code.
struct LaneLayout {  // k_inflate_tokens: one segment per LANE (KT_LANES of these per workgroup)
  static constexpr uint32_t kFastL = 9, kFastD = 7;
  static constexpr uint32_t kOffFastL = 0, kOffSymL = 1024, kOffCntL = 1600, kOffFastD = 1632, kOffSymD = 1888,
                            kOffCntD = 1952, kOffLens = 1984;
  static constexpr uint32_t kBytes = 2308;  // 577 dwords: an odd stride spreads the lanes over the LDS banks
};
struct SharedLayout {  // k_inflate_tokens_sub: one segment per WAVE, its 32 region lanes share the tables
  static constexpr uint32_t kFastL = 9, kFastD = 7;
  static constexpr uint32_t kOffFastL = 0, kOffSymL = 1024, kOffCntL = 1600, kOffFastD = 1632, kOffSymD = 1888,
                            kOffCntD = 1952, kOffLens = 1984;
  static constexpr uint32_t kBytes = 2304;
};
constexpr uint32_t kOffClLut = 32;  // u8[128] inside the (not yet built) literal/length fast table

struct SegmentResult {
  uint32_t status;   // one of the values above
  uint32_t ntok;     // tokens written (raw == 0)
  uint32_t raw;      // 1: the segment is one stored block holding all `out_n` bytes at byte `raw_off`
  uint64_t raw_off;  // of the stream buffer
};

SF_HD uint16_t ld16(const uint8_t* m, uint32_t off) { return *reinterpret_cast<const uint16_t*>(m + off); }
SF_HD void st16(uint8_t* m, uint32_t off, uint32_t v) { *reinterpret_cast<uint16_t*>(m + off) = (uint16_t)v; }

// RFC 1951 3.2.5 (src/decompress.cpp:53-84)
SF_HD void length_info(uint32_t sym /*257..285*/, uint32_t& base, uint32_t& extra) {
  const uint32_t k = sym - 257;
  if (k < 8) { base = 3 + k; extra = 0; return; }
  if (k == 28) { base = 258; extra = 0; return; }
  extra = (k - 4) >> 2;
  base = 3 + ((4 + (k & 3)) << extra);
}
SF_HD void distance_info(uint32_t sym /*0..29*/, uint32_t& base, uint32_t& extra) {
  if (sym < 4) { base = 1 + sym; extra = 0; return; }
  extra = (sym - 2) >> 1;
  base = 1 + ((2 + (sym & 1)) << extra);
}

// Bit positions inside a segment are 32-bit: a segment yields at most 32 KiB, so its bits are capped at 2^31
// (anything longer is padding or garbage and ends in a status either way).
struct BitReader {
  const uint32_t* seg32;  // dword holding the segment's first byte (the buffer is 4-byte aligned and readable
                          // up to the next multiple of 4 bytes)
  uint32_t last;          // index, from seg32, of the last dword of the buffer: nothing past it is read
  uint32_t word;          // index of the next dword to load
  uint32_t first_byte;    // offset of the segment's first byte inside seg32[0]
  uint64_t buf;
  uint32_t cnt;
  uint32_t next;          // dword `word - 1`, loaded one refill ahead of its use
  uint32_t bitpos;        // bits consumed since the segment began
  uint32_t nbits;         // bits the segment holds

  SF_HD void open(const uint8_t* base, uint64_t src_n /* > 0 */, uint64_t seg_begin, uint64_t seg_end) {
    const uint64_t w0 = seg_begin >> 2, wl = (src_n - 1) >> 2;
    seg32 = reinterpret_cast<const uint32_t*>(base) + w0;
    last = wl - w0 < 0xFFFFFFF0ull ? (uint32_t)(wl - w0) : 0xFFFFFFF0u;
    first_byte = (uint32_t)(seg_begin & 3);
    const uint64_t nb = 8 * (seg_end - seg_begin);
    nbits = nb < 0x80000000ull ? (uint32_t)nb : 0x80000000u;
    bitpos = 0;
    seek(0);
  }
  // Branch-free on purpose: the refill loads one dword ahead of its use, and a branch around the load would
  // make the compiler wait for it on the spot (a full memory round trip per refill).  Past the end the last
  // dword repeats; only a truncated or corrupt stream gets that far, and it ends in an error status.
  SF_HD uint32_t load_word(uint32_t w) const { return seg32[w < last ? w : last]; }
  SF_HD void seek(uint32_t byte) {  // byte offset from the segment's first byte; bit accounting is left alone
    const uint32_t b = byte + first_byte;
    word = b >> 2;
    const uint32_t sh = 8 * (b & 3);
    buf = (uint64_t)(load_word(word++) >> sh);
    cnt = 32 - sh;
    next = load_word(word++);
  }
  SF_HD void refill() {  // afterwards at least 33 bits are buffered
    if (cnt <= 32) {
      buf |= (uint64_t)next << cnt;
      cnt += 32;
      next = load_word(word++);
    }
  }
  SF_HD uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }  // n <= 31
  SF_HD void drop(uint32_t n) {
    buf >>= n;
    cnt -= n;
    bitpos += n;
  }
  SF_HD uint32_t get(uint32_t n) {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  // `a` bits of a code, then a `b`-bit field: one shift of the buffer instead of two
  SF_HD uint32_t drop_get(uint32_t a, uint32_t b) {
    const uint32_t v = ((uint32_t)(buf >> a)) & ((1u << b) - 1u);
    drop(a + b);
    return v;
  }
  SF_HD bool overrun() const { return bitpos > nbits; }
};

SF_HD uint32_t bit_reverse(uint32_t code, uint32_t len) {
  uint32_t r = 0;
  for (uint32_t k = 0; k < len; ++k) r |= ((code >> k) & 1u) << (len - 1 - k);
  return r;
}

// Canonical code (huffman/src/table.hpp:177-216) from the code lengths at L::kOffLens: per-length counts,
// symbols sorted by (length, symbol), and the one-read table for codes up to the fast width.  Serial (one
// lane); k_inflate_tokens_sub has a wave-cooperative version of the same tables.
template <class L, bool WIDE>
SF_HD void build_tables(uint8_t* m) {
  constexpr uint32_t off_cnt = WIDE ? L::kOffCntL : L::kOffCntD, off_sym = WIDE ? L::kOffSymL : L::kOffSymD;
  constexpr uint32_t off_fast = WIDE ? L::kOffFastL : L::kOffFastD, fast_bits = WIDE ? L::kFastL : L::kFastD;
  constexpr uint32_t n = WIDE ? 288 : 32;
  const uint8_t* lens = m + L::kOffLens + (WIDE ? 0 : 288);
  for (uint32_t l = 0; l < 16; ++l) st16(m, off_cnt + 2 * l, 0);
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t l = lens[s] & 15u;
    st16(m, off_cnt + 2 * l, ld16(m, off_cnt + 2 * l) + 1u);
  }
  st16(m, off_cnt, 0);  // length 0 = unused symbol
  uint32_t offs[16];
  offs[0] = 0;
  offs[1] = 0;
#pragma unroll
  for (uint32_t l = 1; l < 15; ++l) offs[l + 1] = offs[l] + ld16(m, off_cnt + 2 * l);
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t l = lens[s] & 15u;
    if (!l) continue;
    // offs[l]++ without indexing a register array by a runtime value
    uint32_t at = 0;
#pragma unroll
    for (uint32_t k = 1; k < 16; ++k)
      if (k == l) { at = offs[k]; offs[k] = at + 1; }
    st16(m, off_sym + 2 * at, s);
  }
  constexpr uint32_t fast_n = 1u << fast_bits;
  for (uint32_t k = 0; k < fast_n; ++k) st16(m, off_fast + 2 * k, 0);
  uint32_t code = 0, idx = 0;
  for (uint32_t l = 1; l <= fast_bits; ++l) {
    const uint32_t c = ld16(m, off_cnt + 2 * l);
    for (uint32_t j = 0; j < c; ++j, ++idx, ++code) {
      const uint32_t sym = ld16(m, off_sym + 2 * idx);
      const uint32_t rev = bit_reverse(code & ((1u << l) - 1u), l);  // an over-subscribed code cannot index past the table
      for (uint32_t e = rev; e < fast_n; e += 1u << l) st16(m, off_fast + 2 * e, (sym << 4) | l);
    }
    code <<= 1;
  }
}

// One symbol.  Returns its code length (0: no code matches the next bits) and the symbol in `sym`.
// The caller has refilled: >= 15 bits are buffered (bits past the end of the input read as zero).
template <class L, bool WIDE>
SF_HD uint32_t decode_symbol(const uint8_t* m, const BitReader& br, uint32_t& sym) {
  constexpr uint32_t fast_bits = WIDE ? L::kFastL : L::kFastD, off_fast = WIDE ? L::kOffFastL : L::kOffFastD;
  const uint32_t e = ld16(m, off_fast + 2 * br.peek(fast_bits));
  if (e) {
    sym = e >> 4;
    return e & 15u;
  }
  // longer code: canonical bit-serial walk (one table row per length), rare
  constexpr uint32_t off_cnt = WIDE ? L::kOffCntL : L::kOffCntD, off_sym = WIDE ? L::kOffSymL : L::kOffSymD;
  uint32_t bits = (uint32_t)br.buf, code = 0, first = 0, index = 0;
  for (uint32_t l = 1; l <= 15; ++l) {  // unrolled on the GPU: measured 10 % faster than the rolled loop
    code |= bits & 1u;
    bits >>= 1;
    const uint32_t c = ld16(m, off_cnt + 2 * l);
    if (code < first + c) {
      sym = ld16(m, off_sym + 2 * (index + (code - first)));
      return l;
    }
    index += c;
    first = (first + c) << 1;
    code <<= 1;
  }
  return 0;
}

// Token sinks.  Aligned: groups four tokens into one 16-byte store (the lane-per-segment kernel's lanes write
// 128 KiB apart).  Plain: one dword per token (region lanes start at arbitrary token indices).
struct alignas(16) Tok4 {
  uint32_t a, b, c, d;
};
struct TokenSink {
  uint32_t* out;  // 16-byte aligned, room for 32768 tokens
  uint32_t n;
  uint32_t q0, q1, q2;
  SF_HD void put(uint32_t t) {
    const uint32_t k = n & 3u;
    if (k == 0) q0 = t;
    else if (k == 1) q1 = t;
    else if (k == 2) q2 = t;
    else *reinterpret_cast<Tok4*>(out + (n & ~3u)) = Tok4{q0, q1, q2, t};  // one dwordx4 store
    ++n;
  }
  SF_HD void tick() {}
  SF_HD void flush() {
    const uint32_t k = n & 3u;
    uint32_t* p = out + (n & ~3u);
    if (k > 0) p[0] = q0;
    if (k > 1) p[1] = q1;
    if (k > 2) p[2] = q2;
  }
};
struct PlainSink {
  uint32_t* out;
  uint32_t n;
  SF_HD void put(uint32_t t) { out[n++] = t; }
  SF_HD void tick() {}
  SF_HD void flush() {}
};
// Region lanes: tokens wait in an 8-entry LDS buffer of the lane and leave for global memory every 8th loop
// iteration, all lanes at once.  (On CDNA loads and stores share one counter, vmcnt: with a token store in
// flight every refill of the bit buffer would wait a full memory round trip.)
struct BufferedSink {
  uint32_t* out;
  uint32_t* buf;  // 8 dwords, this lane's
  uint32_t n, flushed, iter;
  SF_HD void put(uint32_t t) { buf[n++ - flushed] = t; }
  SF_HD void tick() {
    if ((++iter & 7u) == 0) flush();
  }
  SF_HD void flush() {
    _Pragma("nounroll") for (uint32_t k = flushed; k < n; ++k) out[k] = buf[k - flushed];
    flushed = n;
  }
};

// Code lengths of a fixed (type 1, src/decompress.cpp:25-40) or dynamic (type 2, :253-367) block into
// L::kOffLens; for type 2 the header is read from `br`.  Returns a status.
template <class L>
SF_HD uint32_t read_lengths(BitReader& br, uint8_t* m, uint32_t type) {
  uint8_t* lens = m + L::kOffLens;
  if (type == 1) {
    _Pragma("nounroll") for (uint32_t s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
    _Pragma("nounroll") for (uint32_t s = 0; s < 32; ++s) lens[288 + s] = 5;
    return kOk;
  }
  br.refill();
  if (br.bitpos + 14 > br.nbits) return kSrcTooSmall;
  const uint32_t hlit = br.get(5) + 257, hdist = br.get(5) + 1, hclen = br.get(4) + 4;
  if (br.bitpos + 3ull * hclen > br.nbits) return kSrcTooSmall;
  uint64_t clp = 0;  // the 19 code-length code lengths, 3 bits each
  _Pragma("nounroll") for (uint32_t k = 0; k < hclen; ++k) {
    br.refill();
    // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (src/decompress.cpp:250-251)
    const uint32_t order = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 8 + ((k - 4) >> 1);
    clp |= (uint64_t)br.get(3) << (3 * order);
  }
  // Lengths that claim more code space than there is (Kraft sum > 1) describe no prefix code: the reference's
  // canonicalize() then makes a code whose value does not fit its bitsize and asserts (huffman/src/code.hpp:33-44).
  // Every decoder of this repository (this one, the wave-parallel tables built from the same lengths, the oracle,
  // the C++ host API) answers kError
  {
    uint32_t kraft = 0;
    _Pragma("nounroll") for (uint32_t s = 0; s < 19; ++s) {
      const uint32_t l = (uint32_t)(clp >> (3 * s)) & 7u;
      kraft += l ? 128u >> l : 0u;
    }
    if (kraft > 128u) return kError;
  }
  // 7-bit lookup table of the code-length code: (symbol << 3) | code bits, 0 = no code
  uint8_t* lut = m + L::kOffFastL + kOffClLut;
  _Pragma("nounroll") for (uint32_t e = 0; e < 128; ++e) lut[e] = 0;
  {
    uint32_t code = 0;
    _Pragma("nounroll") for (uint32_t l = 1; l <= 7; ++l) {
      _Pragma("nounroll") for (uint32_t s = 0; s < 19; ++s)
        if (((clp >> (3 * s)) & 7u) == l) {
          const uint32_t rev = bit_reverse(code & ((1u << l) - 1u), l);
          _Pragma("nounroll") for (uint32_t e = rev; e < 128; e += 1u << l) lut[e] = (uint8_t)((s << 3) | l);
          ++code;
        }
      code <<= 1;
    }
  }
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0;
  while (i < total) {
    br.refill();
    if (br.overrun()) return kSrcTooSmall;
    const uint32_t e = lut[br.peek(7)];
    if (e == 0) return kError;
    br.drop(e & 7u);
    const uint32_t sym = e >> 3;
    if (sym < 16) {
      lens[i++] = (uint8_t)sym;
      prev = sym;
    } else {
      uint32_t rep, val = 0;
      if (sym == 16) {
        if (i == 0) return kError;  // nothing to repeat (src/decompress.cpp:278)
        val = prev;
        rep = 3 + br.get(2);
      } else if (sym == 17) {
        rep = 3 + br.get(3);
      } else {
        rep = 11 + br.get(7);
      }
      if (i + rep > total) return kError;  // run past the last length
      _Pragma("nounroll") for (uint32_t k = 0; k < rep; ++k) lens[i++] = (uint8_t)val;
      prev = val;
    }
  }
  if (br.overrun()) return kSrcTooSmall;
  {
    uint32_t kl = 0, kd = 0;  // Kraft sums in units of 2^-15
    _Pragma("nounroll") for (uint32_t s = 0; s < hlit; ++s) kl += lens[s] ? 32768u >> lens[s] : 0u;
    _Pragma("nounroll") for (uint32_t s = 0; s < hdist; ++s) kd += lens[hlit + s] ? 32768u >> lens[hlit + s] : 0u;
    if (kl > 32768u || kd > 32768u) return kError;
  }
  // move the distance lengths to their fixed place [288..) so both layouts look alike (downwards: the
  // ranges may overlap and the destination is the higher one)
  if (hlit < 288) {
    _Pragma("nounroll") for (uint32_t s = hdist; s-- > 0;) lens[288 + s] = lens[hlit + s];
    _Pragma("nounroll") for (uint32_t s = hlit; s < 288; ++s) lens[s] = 0;
  }
  _Pragma("nounroll") for (uint32_t s = hdist; s < 32; ++s) lens[288 + s] = 0;
  return kOk;
}

template <class L>
SF_HD uint32_t read_tables(BitReader& br, uint8_t* m, uint32_t type) {
  const uint32_t st = read_lengths<L>(br, m, type);
  if (st != kOk) return st;
  build_tables<L, true>(m);
  build_tables<L, false>(m);
  return kOk;
}

// Symbol loop (src/decompress.cpp:122-187) until the end-of-block code or, if sooner, until `end_bit` bits of
// the segment are consumed (a region lane stops where the next region starts).  out_pos: bytes of the segment
// produced before / after; out_limit: where this caller's output must end at the latest; hist: output bytes that
// precede the segment and that its matches may reach (0 for an independent segment; the earlier segments of the
// same strip otherwise -- the reference's one rule is distance <= bytes written, src/decompress.cpp:178).
template <class L, class Sink>
SF_HD uint32_t decode_symbols(BitReader& br, const uint8_t* m, Sink& sink, uint32_t& out_pos, uint32_t out_limit,
                              uint32_t end_bit, bool& hit_eob, uint32_t hist) {
  hit_eob = false;
  while (br.bitpos < end_bit) {
    sink.tick();
    br.refill();
    uint32_t sym;
    const uint32_t l = decode_symbol<L, true>(m, br, sym);
    if (l == 0) return kInvalidLitOrLen;
    if (sym < 256) {
      br.drop(l);
      if (out_pos >= out_limit) return kDstTooSmall;
      sink.put(sym);
      ++out_pos;
      continue;
    }
    if (sym == 256) {
      br.drop(l);
      hit_eob = true;
      break;
    }
    if (sym > 285) return kInvalidLitOrLen;
    uint32_t lbase, lextra;
    length_info(sym, lbase, lextra);
    const uint32_t len = lbase + br.drop_get(l, lextra);
    br.refill();
    uint32_t dsym;
    const uint32_t dl = decode_symbol<L, false>(m, br, dsym);
    if (dl == 0 || dsym > 29) return kInvalidDistance;
    uint32_t dbase, dextra;
    distance_info(dsym, dbase, dextra);
    const uint32_t dist = dbase + br.drop_get(dl, dextra);
    if (dist > out_pos + hist) return kInvalidDistance;  // src/decompress.cpp:178
    if (len > out_limit - out_pos) return kDstTooSmall;
    sink.put(kTokMatchBit | ((len - 3) << 16) | (dist - 1));
    out_pos += len;
    if (br.overrun()) return kSrcTooSmall;
  }
  return br.overrun() ? (uint32_t)kSrcTooSmall : (uint32_t)kOk;
}

// Decodes the blocks of one segment: stream bytes [seg_begin, seg_end) of `src`, which must produce exactly
// out_n (<= 32768) bytes.  Tokens go to tokens[0..ntok).  `m`: LaneLayout::kBytes of scratch.  hist: see decode_symbols.
SF_HD SegmentResult decode_segment(const uint8_t* src, uint64_t src_n, uint64_t seg_begin, uint64_t seg_end,
                                   uint32_t out_n, uint32_t* tokens, uint8_t* m, uint32_t hist = 0) {
  SegmentResult r{kOk, 0, 0, 0};
  if (seg_begin > seg_end || seg_end > src_n) {
    r.status = kSrcTooSmall;
    return r;
  }
  if (seg_begin == seg_end) {
    r.status = kInvalidBlockHeader;  // no header bits at all (src/decompress.cpp:370-373)
    return r;
  }
  BitReader br;
  br.open(src, src_n, seg_begin, seg_end);
  TokenSink sink{tokens, 0, 0, 0, 0};
  uint32_t out_pos = 0;
  uint32_t status = kOk;
  bool last = false;
  while (!last && status == kOk) {
    if (br.bitpos + 3 > br.nbits) {
      // out of input: fine when the segment is complete (a non-final shard ends without BFINAL), else truncated
      if (out_pos != out_n || br.bitpos == 0) status = br.bitpos == 0 ? kInvalidBlockHeader : kSrcTooSmall;
      break;
    }
    br.refill();
    last = br.get(1) != 0;
    const uint32_t type = br.get(2);
    if (type == 3) { status = kInvalidBlockHeader; break; }
    if (type == 0) {
      // stored: src/decompress.cpp:416-436
      br.drop((uint32_t)((8 - (br.bitpos & 7)) & 7));
      br.refill();
      if (br.bitpos + 32 > br.nbits) { status = kSrcTooSmall; break; }
      const uint32_t len = br.get(16);
      br.refill();
      const uint32_t nlen = br.get(16);
      if ((len ^ nlen) != 0xFFFFu) { status = kNoCompressionLenMismatch; break; }
      if (br.bitpos + 8 * len > br.nbits) { status = kSrcTooSmall; break; }
      if (len > out_n - out_pos) { status = kDstTooSmall; break; }
      const uint32_t data_rel = br.bitpos >> 3;
      const uint64_t data_at = seg_begin + data_rel;
      if (sink.n == 0 && out_pos == 0 && len == out_n && len != 0) {
        r.raw = 1;  // the whole segment is this block: the byte-copy kernel takes it from the stream
        r.raw_off = data_at;
        out_pos = len;
      } else {
        for (uint32_t k = 0; k < len; ++k) sink.put(src[data_at + k]);
        out_pos += len;
      }
      br.bitpos += 8 * len;
      br.seek(data_rel + len);
      continue;
    }
    status = read_tables<LaneLayout>(br, m, type);
    if (status != kOk) break;
    bool eob;
    status = decode_symbols<LaneLayout>(br, m, sink, out_pos, out_n, ~0u, eob, hist);
  }
  if (status == kOk && out_pos != out_n) status = kSrcTooSmall;  // the index promised more bytes
  if (status == kOk && r.raw && sink.n != 0) status = kDstTooSmall;
  sink.flush();
  r.status = status;
  r.ntok = sink.n;
  return r;
}

// ---- sub-indexed segments (streams of this library: one block per segment, 32 parse regions of 1024 bytes
// ---- whose first token codes are located by the sub-index) ----

// First lane of a segment: block header and code lengths (build: and the tables, serially).  raw != 0: stored
// segment (bytes at raw_off).  hdr_end: bit offset, from the segment's first byte, of the first token code.
template <class L>
SF_HD uint32_t open_segment(const uint8_t* src, uint64_t src_n, uint64_t seg_begin, uint64_t seg_end, uint32_t out_n,
                            uint8_t* m, bool build, uint32_t& raw, uint64_t& raw_off, uint64_t& hdr_end) {
  raw = 0;
  raw_off = 0;
  hdr_end = 0;
  if (seg_begin > seg_end || seg_end > src_n) return kSrcTooSmall;
  if (seg_begin == seg_end) return kInvalidBlockHeader;
  BitReader br;
  br.open(src, src_n, seg_begin, seg_end);
  if (br.nbits < 3) return kInvalidBlockHeader;
  br.refill();
  br.drop(1);  // BFINAL: the segment ends with its block either way
  const uint32_t type = br.get(2);
  if (type == 3) return kInvalidBlockHeader;
  if (type == 0) {
    br.drop(5);
    if (br.nbits < 40) return kSrcTooSmall;
    const uint32_t len = br.get(16);
    br.refill();
    const uint32_t nlen = br.get(16);
    if ((len ^ nlen) != 0xFFFFu) return kNoCompressionLenMismatch;
    if (40 + 8 * len > br.nbits) return kSrcTooSmall;
    if (len != out_n) return len > out_n ? kDstTooSmall : kSrcTooSmall;
    raw = 1;
    raw_off = seg_begin + 5;
    return kOk;
  }
  const uint32_t st = build ? read_tables<L>(br, m, type) : read_lengths<L>(br, m, type);
  hdr_end = br.bitpos;
  return st;
}

// One region lane: token codes from bit `bit_begin` of the segment up to `bit_end` (or to the end-of-block code
// when until_eob), which must produce exactly the bytes [out_begin, out_end) of the segment.
template <class L>
SF_HD uint32_t decode_region(const uint8_t* src, uint64_t src_n, uint64_t seg_begin, uint64_t seg_end, uint32_t bit_begin,
                             uint32_t bit_end, bool until_eob, uint32_t out_begin, uint32_t out_end, uint32_t* tokens,
                             const uint8_t* m, uint32_t& ntok, uint32_t* lane_buf = nullptr, uint32_t hist = 0) {
  ntok = 0;
  if (seg_begin >= seg_end || seg_end > src_n) return kError;
  BitReader br;
  br.open(src, src_n, seg_begin, seg_end);
  if (bit_begin > br.nbits || (!until_eob && (bit_end < bit_begin || bit_end > br.nbits))) return kError;
  br.seek(bit_begin >> 3);
  br.bitpos = bit_begin & ~7u;
  br.refill();
  br.drop(bit_begin & 7u);
  uint32_t out_pos = out_begin;
  bool eob;
  uint32_t st;
  if (lane_buf) {
    BufferedSink sink{tokens, lane_buf, 0, 0, 0};
    st = decode_symbols<L>(br, m, sink, out_pos, out_end, until_eob ? ~0u : bit_end, eob, hist);
    sink.flush();
    ntok = sink.n;
  } else {
    PlainSink sink{tokens, 0};
    st = decode_symbols<L>(br, m, sink, out_pos, out_end, until_eob ? ~0u : bit_end, eob, hist);
    ntok = sink.n;
  }
  if (st != kOk) return st;
  if (out_pos != out_end) return kError;                    // the sub-index and the stream disagree
  if (until_eob ? !eob : (eob || br.bitpos != bit_end)) return kError;
  return kOk;
}

}  // namespace inflate
}  // namespace sf
