// sf_inflate.hip -- GPU decode of block-indexed DEFLATE streams: the reference's
// decompress() (/root/reference/src/decompress.cpp:402-461) for streams whose byte-aligned segments and
// independently decodable strips are known -- every stream this library writes (one segment per 32 KiB chunk;
// no match reaches before its strip of block_bytes; the index is the chunk offset table k_scan already
// produces), and e.g. zlib streams with Z_SYNC_FLUSH every 32 KiB and Z_FULL_FLUSH at the strip boundaries.
// Arbitrary DEFLATE is serial (README.md:5-6); the index is what makes it parallel.
//
//   k_inflate_tokens_sub   the bit-serial half with the region sub-index this library's compressor writes: 32 lanes per
//                     segment, one per 1024 bytes of output, 64 bit streams decoded side by side in lockstep.
//   k_inflate_tokens_spec  the same wave with the segment index only: the lanes find their token boundaries themselves
//                     (a decoder started on a wrong bit falls in with the true token chain within a few hundred bits),
//                     count, and the spans they settled on then serve as the sub-index.
//   k_inflate_tokens  behind k_inflate_tokens_spec, for the segments that are not one clean block: Huffman decoding cannot
//                     be split inside a segment in general, so one LANE per segment runs decode_segment()
//                     (sf_inflate_core.h) with its own code tables in a 2,308-byte slice of LDS and its own 64-bit bit
//                     buffer fed by dword loads one refill ahead.  Output of all three: the k_lz77 token format.
//   k_inflate_bytes   the byte-copy half (src/decompress.cpp:157-187,388-398), one 512-thread workgroup per
//                     strip, its segments in order, the output window a 36 KiB ring in LDS (the 32 KiB a match
//                     may reach back + the step in flight), steps of <= 1024 tokens / 3968 bytes: a
//                     workgroup prefix sum places the tokens, every thread then takes BYTES (token found by
//                     bitmap + popcount): sources before the step are final and copied at once, sources
//                     inside it become 16-bit pointers that pointer jumping resolves in a few barrier-
//                     separated rounds, whatever the nesting or overlap.  A segment that is one stored block
//                     is copied straight from the stream.
//   k_inflate_status  first non-zero segment status in stream order = what the serial decoder would report.
#include "sf_device.h"

#include <stdlib.h>
#include "sf_inflate_core.h"

namespace sf {

namespace {

// k_inflate_tokens: segments (lanes) per workgroup.  A lane's decode is one long dependent chain and the CU holds the tables
// of 64 lanes at most (LDS), so what counts is how many waves those lanes are spread over: 4 lanes per wave = 16 waves
// per CU measured best (1 GiB of text: 23.4 ms; 64 lanes in one wave 28.8, 32: 24.9, 16: 27.2, 8: 24.4, 2: 29.7, 1: 56.4)
constexpr uint32_t KT_LANES = 4;
constexpr uint32_t kRetrySerial = 0xFFFFFFFEu;  // SegInfo.status between k_inflate_tokens_spec and k_inflate_tokens: not decoded yet
constexpr uint32_t KT_LDS = KT_LANES * inflate::LaneLayout::kBytes;
constexpr uint32_t KB_THREADS = 512;
constexpr uint32_t KB_TPT = 3;      // tokens per thread per step (2: 4.5 ms per GiB, 3: 3.75, 4: scratch)
constexpr uint32_t KB_SPAN = 3968;  // output bytes per step (pointer array)
constexpr uint32_t KB_AUX = 2 * KB_SPAN + 4 * KB_THREADS * KB_TPT + 8 * (KB_SPAN / 32) + 64;
constexpr uint32_t KB_RING = kWindow + 4096;             // the window + the step in flight (KB_SPAN bytes at most)
constexpr uint32_t KB_LDS = KB_RING + KB_AUX;            // 52,000 B: three workgroups per CU
static_assert(3 * KB_LDS <= 160 * 1024, "k_inflate_bytes: three workgroups per CU");
static_assert(KB_SPAN <= KB_RING - kWindow && KB_RING % 16 == 0 && kChunk % 4 == 0, "ring geometry");

__global__ __launch_bounds__(KT_LANES) void k_inflate_tokens(const uint8_t* __restrict__ src, uint64_t src_n,
                                                            const uint64_t* __restrict__ index, uint32_t nseg,
                                                            uint64_t dst_n, uint32_t* __restrict__ tokens,
                                                            SegInfo* __restrict__ info, uint32_t sps,
                                                            uint32_t only_retry) {
  extern __shared__ __align__(16) uint8_t s_tables[];
  const uint32_t seg = blockIdx.x * KT_LANES + threadIdx.x;
  if (seg >= nseg) return;
  if (only_retry && info[seg].status != kRetrySerial) return;  // behind k_inflate_tokens_spec: what that one left
  const uint64_t lo = index[seg], hi = index[seg + 1];
  const uint64_t obase = (uint64_t)seg * kChunk;
  const uint32_t out_n = dst_n > obase ? (uint32_t)(dst_n - obase < kChunk ? dst_n - obase : kChunk) : 0u;
  const inflate::SegmentResult r = inflate::decode_segment(src, src_n, lo, hi, out_n, tokens + (uint64_t)seg * kChunk,
                                                           s_tables + threadIdx.x * inflate::LaneLayout::kBytes,
                                                           (seg % sps) * kChunk);
  SegInfo si;
  si.status = r.status;
  si.ntok = r.ntok;
  si.raw = (r.raw ? kSegRaw : 0u) | kSegSerial;
  si.out_n = out_n;
  si.raw_off = r.raw_off;
  info[seg] = si;
}

// Wave-cooperative version of inflate::build_tables for the SharedLayout: same counts, sorted symbols and
// one-read tables, built by 64 lanes with ballots instead of by one lane with loops.
template <bool WIDE>
__device__ __forceinline__ void build_tables_wave(uint8_t* m, uint32_t lane) {
  using L = inflate::SharedLayout;
  constexpr uint32_t off_cnt = WIDE ? L::kOffCntL : L::kOffCntD, off_sym = WIDE ? L::kOffSymL : L::kOffSymD;
  constexpr uint32_t off_fast = WIDE ? L::kOffFastL : L::kOffFastD, fast_bits = WIDE ? L::kFastL : L::kFastD;
  constexpr uint32_t n = WIDE ? 288 : 32, nblk = (n + 63) / 64, fast_n = 1u << fast_bits;
  const uint8_t* lens = m + L::kOffLens + (WIDE ? 0 : 288);
  const uint64_t lt_mask = (1ull << lane) - 1;
  uint32_t myl[nblk];
#pragma unroll
  for (uint32_t b = 0; b < nblk; ++b) {
    const uint32_t s = b * 64 + lane;
    myl[b] = s < n ? (lens[s] & 15u) : 0u;
  }
  {
    uint32_t* f32 = reinterpret_cast<uint32_t*>(m + off_fast);
#pragma nounroll
    for (uint32_t k = lane; k < fast_n / 2; k += 64) f32[k] = 0;
  }
  __syncthreads();  // the code lengths are read, the fast table (which lent its start to the header parse) is clear
  uint32_t code = 0, index = 0;  // uniform: first code / first sorted slot of the current length
#pragma nounroll
  for (uint32_t l = 1; l <= 15; ++l) {
    uint32_t cnt_l = 0;
#pragma unroll
    for (uint32_t b = 0; b < nblk; ++b) {
      const uint64_t mask = __ballot(myl[b] == l);
      if (myl[b] == l) {
        const uint32_t rank = cnt_l + (uint32_t)__popcll(mask & lt_mask);  // symbols of one length in symbol order
        const uint32_t s = b * 64 + lane;
        inflate::st16(m, off_sym + 2 * (index + rank), s);
        if (l <= fast_bits) {
          const uint32_t c = (code + rank) & ((1u << l) - 1u);  // an over-subscribed code cannot index past the table
          const uint32_t rev = __builtin_bitreverse32(c) >> (32 - l);
#pragma nounroll
          for (uint32_t e = rev; e < fast_n; e += 1u << l) inflate::st16(m, off_fast + 2 * e, (s << 4) | l);
        }
      }
      cnt_l += (uint32_t)__popcll(mask);
    }
    if (lane == 0) inflate::st16(m, off_cnt + 2 * l, cnt_l);
    index += cnt_l;
    code = (code + cnt_l) << 1;
  }
  if (lane == 0) inflate::st16(m, off_cnt, 0);
}

// ---- the token loop of k_inflate_tokens_sub ----
// The 64 region lanes of a wave in lockstep, one token per lane and iteration, written for what bounds this kernel:
// instruction issue (vector and, with one scalar unit per CU, scalar) and vmcnt, the one counter loads and stores share.
//   * straight-line: literal, length and distance are decoded by every lane in every iteration and selected
//     afterwards; base / extra-bit pairs come from a 64-entry table in LDS (RFC 1951 3.2.5).  A code longer than the
//     one-read tables is not walked bit by bit (some lane of 64 has one in most iterations): canonical codes of
//     length l, left-justified to 15 bits, lie below limit[l] = (first[l] + count[l]) << (15 - l), the limits never
//     decrease, so the length is 1 + the number of limits the next 15 bits reach -- a handful of compares.
//   * no bit buffer: a lane keeps its bit position; the next 32 bits are two dwords of its STREAM WINDOW in LDS and
//     one v_alignbit.  The window (kWinDwords dwords from the lane's position) is loaded once per PERIOD of kPeriod
//     iterations into registers and moved to LDS at the next period's start, so the load has a whole period to
//     arrive.  (A dword load per lane "one refill ahead" sounds the same, but lanes refill in different iterations:
//     the wave then waits in nearly every iteration for a load issued one iteration earlier.)  A lane that outruns
//     its window reads those dwords from memory.
//   * the period's tokens stay in registers and leave as one 16-byte store per lane, issued in the same burst as the
//     window loads and, like them, waited for a period later.
constexpr uint32_t kPeriod = 4, kWinDwords = 8;
constexpr uint32_t kWinRow = kWinDwords + 1;  // an odd row length spreads the lanes' rows over the LDS banks
struct RegionLds {
  uint32_t win[64][kWinRow];  // a lane's window: consecutive dwords, so that two of them are one ds_read2
  uint32_t lut[64];           // [k]: length symbol 257+k, [32+s]: distance symbol s -- base | extra bits << 16
  uint32_t lim[2][2][16];     // [segment of the wave][0: literal/length, 1: distance][code length]: see above
  int32_t base[2][2][16];     // sorted-symbol index of the first code of a length - that code
};
struct alignas(4) Dwords4 {  // four dwords at a dword-aligned address: the compiler picks the widest legal access
  uint32_t a, b, c, d;
};

// limit[] / base[] of one code from its per-length counts (lane l < 16 writes entry l; count[0] is 0)
template <bool WIDE>
__device__ void build_limits(const uint8_t* m, uint32_t* lim, int32_t* base, uint32_t lane) {
  using L = inflate::SharedLayout;
  constexpr uint32_t off_cnt = WIDE ? L::kOffCntL : L::kOffCntD;
  if (lane < 16) {
    uint32_t first = 0, index = 0, c = 0;
    for (uint32_t l = 1; l <= lane; ++l) {  // first code and first sorted slot of length `lane`
      first = (first + c) << 1;
      index += c;
      c = inflate::ld16(m, off_cnt + 2 * l);
    }
    lim[lane] = lane ? (first + c) << (15 - lane) : 0u;
    base[lane] = (int32_t)index - (int32_t)first;
  }
}

// a code longer than the one-read table: its length and symbol from the next 32 stream bits (0: no code matches)
template <bool WIDE>
__device__ __forceinline__ uint32_t long_code(const uint8_t* m, const uint32_t* lim, const int32_t* base, uint32_t bits,
                                              uint32_t& sym) {
  using L = inflate::SharedLayout;
  constexpr uint32_t fast_bits = WIDE ? L::kFastL : L::kFastD, off_sym = WIDE ? L::kOffSymL : L::kOffSymD;
  const uint32_t c15 = __builtin_bitreverse32(bits) >> 17;  // the first stream bit is the code's most significant
  uint32_t l = fast_bits + 1;
#pragma unroll
  for (uint32_t k = fast_bits + 1; k < 15; ++k) l += c15 >= lim[k] ? 1u : 0u;
  const bool hit = c15 < lim[15];
  const uint32_t idx = (uint32_t)(base[l] + (int32_t)(c15 >> (15 - l)));
  sym = inflate::ld16(m, off_sym + 2 * (hit ? idx : 0u));
  return hit ? l : 0u;
}

struct RegionOut {
  uint32_t st, ntok;
  uint32_t bit;    // where the lane stopped (bit offset from the segment's first byte)
  uint32_t bytes;  // output bytes of its tokens
  bool eob;        // it met the end-of-block code
  // COUNT: a checkpoint -- the first period start at or behind cp_target: its bit offset (0: none), tokens and bytes before it
  uint32_t cp_bit, cp_tok, cp_bytes;
};

// COUNT (k_inflate_tokens_spec's passes before the last): nothing is written and nothing is known about the output
// position -- the lane's tokens are counted from bit_begin to the first token boundary at or behind bit_end, or to the
// end-of-block code, which then stays unread (the lanes behind find it again and stand still on it).
template <class L, bool COUNT>
__device__ void decode_regions_lockstep(const uint8_t* src, uint64_t src_n, uint64_t seg_begin, uint64_t seg_end,
                                        uint32_t bit_begin, uint32_t bit_end, bool until_eob, uint32_t out_begin,
                                        uint32_t out_end, uint32_t* tokens, const uint8_t* m, uint32_t half, uint32_t hist,
                                        bool decode, uint32_t lane, RegionLds& R, RegionOut& out, uint32_t cp_target = ~0u) {
  using namespace inflate;
  uint32_t st = kOk;
  bool active = decode;
  // the segment's dwords: seg32[0] holds its first byte; nothing past seg32[last] is read (BitReader::open)
  const uint32_t* seg32 = reinterpret_cast<const uint32_t*>(src);
  uint32_t last = 0, bias = 0, nbits = 0;  // bias: bits of seg32[0] before the segment's first byte
  if (active) {
    if (seg_begin >= seg_end || seg_end > src_n) {
      st = kError;
      active = false;
    } else {
      const uint64_t w0 = seg_begin >> 2, wl = (src_n - 1) >> 2;
      seg32 += w0;
      last = wl - w0 < 0xFFFFFFF0ull ? (uint32_t)(wl - w0) : 0xFFFFFFF0u;
      bias = 8 * (uint32_t)(seg_begin & 3);
      const uint64_t nb = 8 * (seg_end - seg_begin);
      nbits = nb < 0x80000000ull ? (uint32_t)nb : 0x80000000u;
      if (bit_begin > nbits || (!until_eob && (bit_end < bit_begin || bit_end > nbits))) {
        st = kError;
        active = false;
      }
    }
  }
  // positions below are bit indices from bit 0 of seg32[0]
  uint32_t ab = bias + bit_begin;
  const uint32_t end_ab = until_eob ? ~0u : bias + bit_end, lim_ab = bias + nbits;
  if (active && ab >= end_ab) active = false;  // an empty region: no token
  const bool started = active;
  uint32_t out_pos = out_begin, n = 0;
  bool eob = false;
  const uint32_t* limL = R.lim[half][0];
  const uint32_t* limD = R.lim[half][1];
  const int32_t* baseL = R.base[half][0];
  const int32_t* baseD = R.base[half][1];

  uint32_t wbase = 0, wvalid = 0;  // the window in LDS: first dword index, dwords it holds
  uint32_t pbase = 0, pvalid = 0;  // the one on its way
  bool pending = false;            // ... if one is
  Dwords4 p0{0, 0, 0, 0}, p1{0, 0, 0, 0};
  // A period of four tokens takes 38 bits of a text stream on average, 192 at the very most, and a window holds 256: a
  // lane asks for the next one only once it is kReload dwords into the one it has (round 4; before: every period, 32
  // bytes requested per 5 consumed -- most of k_inflate_tokens_sub's 12 GB of reads per GiB).  A lane that then outruns
  // its window inside a period reads the missing dwords from memory, as it always did.
  constexpr uint32_t kReload = 4;  // 1 / 2 / 3 / 4 / 5 on one box: 3.44 / 2.96 / 2.76 / 2.71 / 2.93 ms per GiB
  auto load_window = [&](bool first) {
    const uint32_t at = ab >> 5;
    pending = active && (first || at - wbase >= kReload);
    if (pending) {
      pbase = at;
      const uint32_t avail = last >= pbase ? last - pbase + 1u : 0u;              // dwords readable from there
      pvalid = avail < kWinDwords ? (avail & ~3u) : kWinDwords;                   // whole 16-byte chunks
      if (pvalid > 0) p0 = *reinterpret_cast<const Dwords4*>(seg32 + pbase);
      if (pvalid > 4) p1 = *reinterpret_cast<const Dwords4*>(seg32 + pbase + 4);
    }
  };
  auto window_to_lds = [&]() {
    if (pending) {
      uint32_t* row = R.win[lane];
      row[0] = p0.a; row[1] = p0.b; row[2] = p0.c; row[3] = p0.d;
      row[4] = p1.a; row[5] = p1.b; row[6] = p1.c; row[7] = p1.d;
      wbase = pbase;
      wvalid = pvalid;
    }
  };
  // the 32 stream bits from bit position a on
  auto peek = [&](uint32_t a) -> uint32_t {
    const uint32_t w = a >> 5, i = w - wbase;
    // the window is read in any case (index clamped) and overridden in the rare other case: written as an if / else
    // the two loads are merged into one FLAT load of a selected pointer, slow and counted by vmcnt
    const bool inside = i + 1 < wvalid;
    const uint32_t i0 = inside ? i : 0u;
    uint32_t d0 = R.win[lane][i0], d1 = R.win[lane][i0 + 1];
    asm volatile("" : "+v"(d0), "+v"(d1));  // pinned: the LDS reads happen here, as LDS reads
    if (!inside) {  // from memory, the last dword repeating past the end (BitReader::load_word)
      d0 = seg32[w < last ? w : last];
      d1 = seg32[w + 1 < last ? w + 1 : last];
      // consumed inside the branch: the wait for these two loads stays in here, instead of a wait at the join that
      // every lane pays in every iteration (and that would also wait for the period's window loads and token store)
      asm volatile("" : "+v"(d0), "+v"(d1));
    }
    return __builtin_amdgcn_alignbit(d1, d0, a & 31);
  };

  uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, made = 0;  // the period's tokens
  uint32_t nflushed = 0;
  // Full periods are kept back (kHold of them) and leave together with the next one: (kHold + 1) x 16 bytes to consecutive
  // addresses at once instead of a 16-byte store per period.  The lanes' token streams are 1.4 KB apart, so every store
  // opens a line of its own, and a line written in eight visits costs the memory side three times its bytes: WRITE_SIZE
  // of this kernel 4.7 GB per GiB of output with kHold = 0, 2.5 GB with 3 (the tokens are 1.6 GB); the time is the same
  // within the noise of the pool (2.68 / 2.74 / 2.87 / 2.87 ms for 0 / 1 / 2 / 3 in one run, 2.80 against 2.60 for 0 / 1 in
  // another) -- kept for the traffic.
  // (Round 5: two held periods, not three: with the four registers that frees the kernel fits 96 VGPRs -- five waves per SIMD
  // instead of four, which is what its 7.7 KB of LDS allow per CU: 2.90 -> 2.79 ms, index-only 4.55 -> 4.33.)
  constexpr uint32_t kHold = 2;
  static_assert(kHold <= 3, "held periods live in named registers");
  Dwords4 h0{0, 0, 0, 0}, h1{0, 0, 0, 0}, h2{0, 0, 0, 0};  // h0: the newest held period
  uint32_t nheld = 0;
  auto store_held = [&]() {  // oldest first
    if (kHold >= 3 && nheld >= 3) { *reinterpret_cast<Dwords4*>(tokens + nflushed) = h2; nflushed += 4; }
    if (kHold >= 2 && nheld >= 2) { *reinterpret_cast<Dwords4*>(tokens + nflushed) = h1; nflushed += 4; }
    if (kHold >= 1 && nheld >= 1) { *reinterpret_cast<Dwords4*>(tokens + nflushed) = h0; nflushed += 4; }
    nheld = 0;
  };
  auto flush = [&]() {
    if (COUNT) return;
    if (made == kPeriod) {
      if (nheld == kHold) {
        store_held();
        *reinterpret_cast<Dwords4*>(tokens + nflushed) = Dwords4{t0, t1, t2, t3};
        nflushed += 4;
      } else {
        h2 = h1; h1 = h0; h0 = Dwords4{t0, t1, t2, t3};
        ++nheld;
      }
    } else {
      store_held();
      if (made) {  // the lane stopped inside the period (once per lane)
        tokens[nflushed] = t0;
        if (made > 1) tokens[nflushed + 1] = t1;
        if (made > 2) tokens[nflushed + 2] = t2;
      }
      nflushed += made;
    }
    made = 0;
  };

  // one token (src/decompress.cpp:122-187; the order of the checks is inflate::decode_symbols')
  auto step = [&](uint32_t& tok) {
    const uint32_t P = peek(ab);
    uint32_t e = ld16(m, L::kOffFastL + 2 * (P & ((1u << L::kFastL) - 1u)));
    if (e == 0) {  // a longer code, or none
      uint32_t sy = 0;
      const uint32_t l = long_code<true>(m, limL, baseL, P, sy);
      e = l ? (sy << 4) | l : 0u;
    }
    const uint32_t l = e & 15u, sym = e >> 4;
    const bool is_lit = sym < 256, is_eob = sym == 256, is_len = sym - 257u < 29u;
    const uint32_t lrec = R.lut[is_len ? sym - 257u : 31u];
    const uint32_t lx = lrec >> 16;
    const uint32_t len = (lrec & 0xFFFFu) + __builtin_amdgcn_ubfe(P, l, lx);
    const uint32_t ab2 = ab + l + lx;
    // the distance, decoded by every lane (only a length symbol's lane keeps it)
    const uint32_t Q = peek(ab2);
    uint32_t de = ld16(m, L::kOffFastD + 2 * (Q & ((1u << L::kFastD) - 1u)));
    if (is_len && de == 0) {
      uint32_t sy = 0;
      const uint32_t dl2 = long_code<false>(m, limD, baseD, Q, sy);
      de = dl2 ? (sy << 4) | dl2 : 0u;
    }
    const uint32_t dl = de & 15u, dsym = de >> 4;
    const uint32_t drec = R.lut[32u + (dsym < 31u ? dsym : 31u)];
    const uint32_t dx = drec >> 16;
    const uint32_t dist = (drec & 0xFFFFu) + __builtin_amdgcn_ubfe(Q, dl, dx);
    const uint32_t ab3 = ab2 + dl + dx;
    // what the serial decoder would report, in its order (selects from the last check to the first, so that the
    // first one that fails wins)
    uint32_t em = ab3 > lim_ab ? (uint32_t)kSrcTooSmall : (uint32_t)kOk;
    em = len > out_end - out_pos ? (uint32_t)kDstTooSmall : em;
    em = (dl == 0 || dsym > 29 || dist > out_pos + hist) ? (uint32_t)kInvalidDistance : em;  // src/decompress.cpp:178
    const uint32_t el = out_pos >= out_end ? (uint32_t)kDstTooSmall : (uint32_t)kOk;
    uint32_t err = is_len ? em : (uint32_t)kInvalidLitOrLen;  // neither literal, end-of-block nor length: sym > 285
    err = is_eob ? (uint32_t)kOk : err;
    err = is_lit ? el : err;
    err = l == 0 ? (uint32_t)kInvalidLitOrLen : err;
    const bool ok = err == kOk;
    const bool emits = ok && !is_eob;
    tok = is_lit ? sym : (kTokMatchBit | ((len - 3) << 16) | (dist - 1));
    out_pos += emits ? (is_lit ? 1u : len) : 0u;
    ab = ok ? (is_len ? ab3 : ((COUNT && is_eob) ? ab : ab + l)) : ab;
    n += emits ? 1u : 0u;
    made += emits ? 1u : 0u;
    eob = eob || (ok && is_eob);
    st = ok ? st : err;
    // (COUNT: no output limit stops a lane that runs off the segment's end on literals -- past the end the last dword
    // repeats for ever; the position does.  Such a lane reports kSrcTooSmall below.)
    active = ok && !is_eob && ab < end_ab && (!COUNT || ab <= lim_ab);
  };

  // the first window: loaded and waited for on the spot
  load_window(true);
  uint32_t cp_ab = 0, cp_tok = 0, cp_bytes = 0;  // COUNT: see RegionOut
  const uint32_t cp_at = (COUNT && cp_target != ~0u) ? bias + cp_target : ~0u;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    if (COUNT) {
      const bool hit = active && cp_ab == 0 && ab >= cp_at;
      cp_ab = hit ? ab : cp_ab;
      cp_tok = hit ? n : cp_tok;
      cp_bytes = hit ? out_pos - out_begin : cp_bytes;
    }
    // period start: last period's window (arrived meanwhile) into LDS, last period's tokens out, next window's loads
    window_to_lds();
    flush();
    load_window(false);
    uint32_t tk;
    if (active) { step(tk); t0 = tk; }
    if (active) { step(tk); t1 = tk; }
    if (active) { step(tk); t2 = tk; }
    if (active) { step(tk); t3 = tk; }
  }
  flush();
  flush();  // (made is 0 now: this one only lets the kept-back periods go)
  out.ntok = n;
  out.bit = ab - bias;
  out.bytes = out_pos - out_begin;
  out.eob = eob;
  out.cp_bit = cp_ab ? cp_ab - bias : 0u;
  out.cp_tok = cp_tok;
  out.cp_bytes = cp_bytes;
  if (COUNT) {
    out.st = (st == kOk && ab > lim_ab) ? (uint32_t)kSrcTooSmall : st;
    return;
  }
  if (started) {
    if (st == kOk) {
      if (ab > lim_ab) st = kSrcTooSmall;
      else if (out_pos != out_end) st = kError;  // the sub-index and the stream disagree
      else if (until_eob ? !eob : (eob || ab != end_ab)) st = kError;
    }
  } else if (decode && st == kOk) {
    // no token at all: the region must be empty in both views (inflate::decode_region on such a region)
    if (out_pos != out_end || until_eob) st = kError;
  }
  out.st = st;
}

// Streams of this library: one block per segment and a sub-index naming, for each of the 32 parse regions
// (1024 bytes of output; k_lz77 never lets a match cross them), the bit offset of the region's first token
// code and the number of tokens before it (k_emit writes both).  One wave takes TWO segments, 32 lanes each:
// lanes 0 and 32 read the two block headers side by side, all 64 lanes build each segment's code tables
// (10-bit literal/length and 8-bit distance one-read tables: a longer code is a rarity), then every lane
// decodes one region, 64 bit streams side by side in lockstep (decode_regions_lockstep above), writing tokens at
// their compact positions.  The kernel is bound by instruction issue, so full waves matter more than
// occupancy.  The sub-index is checked against the stream (first code right after the header, every lane ends
// exactly where the next begins, exact byte and token counts): a wrong sub-index is an error, never wrong output.
//
// k_inflate_tokens_spec: streams with the segment index ONLY.  Where a token starts inside a segment is unknown, but a
// Huffman decoder started at a wrong bit falls in with the true token chain quickly (DEFLATE streams of the text
// workload: half the wrong starts within 42 bits, 99.9 % within 360), and two decoders that stand on the same bit at a
// token boundary stay together for good.  So the segment's code bits are cut into 32 spans of equal length, one per lane:
//   1. look-back: every lane decodes the kSpecLookBack bits before its span, from an arbitrary bit, counting nothing:
//      the first boundary it reaches inside its span is its guess of where the span's first token starts;
//   2. count: every lane decodes its span from that guess to the first boundary inside the next span and counts tokens
//      and bytes.  Lane 0's start is exact (the end of the block header), so every lane whose start equals the end its
//      predecessor reported is exact by induction; a lane whose start differs takes the predecessor's end and decodes
//      again, until none differs (at most 32 rounds: one more lane is final after each);
//   3. the spans are now what a sub-index is to k_inflate_tokens_sub -- first bit, last bit, tokens and bytes before --
//      and its lockstep pass writes the tokens, with every check the serial decoder makes.
// A segment's blocks are taken one after the other, as the serial decoder reads them (tokens_wave_spec below): a stored
// block becomes the raw copy (the first block, holding the whole segment) or literal tokens, a coded block goes through 1-3,
// the empty stored blocks a flush leaves are read on the spot.  A segment is finished here only if everything about it was
// in order; an error of any kind leaves it to k_inflate_tokens (info.status = kRetrySerial, launched behind this kernel for
// those segments only): statuses are the serial decoder's by construction.
constexpr uint32_t kSpecLookBack = 512;
constexpr uint32_t kSpecCheck = 1024;  // the checkpoint of a counting pass: this many bits behind the lane's start

__device__ __forceinline__ void tokens_wave_sub(const uint8_t* __restrict__ src, uint64_t src_n, const uint64_t* __restrict__ index,
                                            const uint32_t* __restrict__ subidx, uint32_t nseg, uint64_t dst_n,
                                            uint32_t* __restrict__ tokens, SegInfo* __restrict__ info, uint32_t sps) {
  using L = inflate::SharedLayout;
  __shared__ __align__(16) uint8_t s_tab[2][L::kBytes];
  __shared__ __align__(16) RegionLds s_reg;
  __shared__ uint32_t s_open[2][2];
  __shared__ uint64_t s_open64[2][2];
  const uint32_t lane = threadIdx.x, half = lane >> 5, hl = lane & 31;
  const uint32_t seg = 2 * blockIdx.x + half;
  const bool live = seg < nseg;
  const uint64_t lo = live ? index[seg] : 0, hi = live ? index[seg + 1] : 0;
  const uint64_t obase = (uint64_t)seg * kChunk;
  const uint32_t out_n = (live && dst_n > obase) ? (uint32_t)(dst_n - obase < kChunk ? dst_n - obase : kChunk) : 0u;
  {
    // RFC 1951 3.2.5 as a table: base | extra bits << 16
    uint32_t base = 0, extra = 0;
    if (lane < 29) inflate::length_info(257 + lane, base, extra);
    else if (lane >= 32 && lane < 62) inflate::distance_info(lane - 32, base, extra);
    s_reg.lut[lane] = base | (extra << 16);
  }
  if (hl == 0) {
    uint32_t raw = 0;
    uint64_t raw_off = 0, hdr_end = 0;
    s_open[half][0] = live ? inflate::open_segment<L>(src, src_n, lo, hi, out_n, s_tab[half], false, raw, raw_off, hdr_end)
                           : (uint32_t)inflate::kError;
    s_open[half][1] = raw;
    s_open64[half][0] = raw_off;
    s_open64[half][1] = hdr_end;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t h = 0; h < 2; ++h) {
    if (s_open[h][0] == inflate::kOk && !s_open[h][1]) {  // uniform
      build_tables_wave<true>(s_tab[h], lane);
      build_tables_wave<false>(s_tab[h], lane);
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t h = 0; h < 2; ++h) {
    if (s_open[h][0] == inflate::kOk && !s_open[h][1]) {  // uniform
      build_limits<true>(s_tab[h], s_reg.lim[h][0], s_reg.base[h][0], lane);
      build_limits<false>(s_tab[h], s_reg.lim[h][1], s_reg.base[h][1], lane);
    }
  }
  __syncthreads();
  uint32_t status = s_open[half][0];
  const uint32_t raw = s_open[half][1];
  uint32_t st = inflate::kOk, n = 0, tok0 = 0;
  const bool decode = live && status == inflate::kOk && !raw;
  uint32_t bit0 = 0, bit1 = 0, tok1 = 0, ob = 0, oe = 0;
  bool go = false;
  if (decode) {
    const uint32_t* sub = subidx + (uint64_t)seg * 2 * kSubRegions;
    bit0 = sub[2 * hl];
    tok0 = sub[2 * hl + 1];
    bit1 = hl + 1 < kSubRegions ? sub[2 * hl + 2] : 0u;
    tok1 = hl + 1 < kSubRegions ? sub[2 * hl + 3] : 0u;
    ob = hl * kSubBytes < out_n ? hl * kSubBytes : out_n;
    oe = (hl + 1) * kSubBytes < out_n ? (hl + 1) * kSubBytes : out_n;
    // (tokens before a region) <= (bytes before it) also bounds the token stores
    if ((hl == 0 && bit0 != s_open64[half][1]) || tok0 > ob) st = inflate::kError;
    else go = true;
  }
  {
    RegionOut o;
    decode_regions_lockstep<L, false>(src, src_n, lo, hi, bit0, bit1, hl + 1 == kSubRegions, ob, oe,
                                      tokens + (uint64_t)seg * kChunk + tok0, s_tab[half], half, (seg % sps) * kChunk, go, lane,
                                      s_reg, o);
    n = o.ntok;
    if (go) {
      st = o.st;
      if (st == inflate::kOk && hl + 1 < kSubRegions && tok0 + n != tok1) st = inflate::kError;
    }
  }
  // per segment: first failing region in stream order, token total from the last region lane
  const uint64_t failed = __ballot(decode && st != inflate::kOk);
#pragma unroll
  for (uint32_t h = 0; h < 2; ++h) {
    const uint64_t fh = (failed >> (32 * h)) & 0xFFFFFFFFull;
    const uint32_t first_st = fh ? (uint32_t)__builtin_amdgcn_readlane(st, __builtin_amdgcn_readfirstlane(__builtin_ctzll(fh)) + 32 * h) : 0u;
    const uint32_t total = __builtin_amdgcn_readlane(tok0 + n, 32 * h + 31);
    if (lane == 32 * h && live) {
      SegInfo si;
      si.status = status != inflate::kOk ? status : (fh ? first_st : (uint32_t)inflate::kOk);
      si.ntok = (status == inflate::kOk && !raw) ? total : 0u;
      si.raw = raw;
      si.out_n = out_n;
      si.raw_off = s_open64[half][0];
      info[seg] = si;
    }
  }
}

__attribute__((amdgpu_waves_per_eu(5, 5)))  // 96 VGPRs: what the workgroup's LDS allows per SIMD
__global__ __launch_bounds__(64, 2) void k_inflate_tokens_sub(const uint8_t* __restrict__ src, uint64_t src_n,
                                                             const uint64_t* __restrict__ index,
                                                             const uint32_t* __restrict__ subidx, uint32_t nseg,
                                                             uint64_t dst_n, uint32_t* __restrict__ tokens,
                                                             SegInfo* __restrict__ info, uint32_t sps) {
  tokens_wave_sub(src, src_n, index, subidx, nseg, dst_n, tokens, info, sps);
}

// ---- k_inflate_tokens_spec: the wave, block after block ----
// header of the block that starts at bit `at` of the segment (one lane); a coded block's code lengths go to m
struct BlockOpen {
  uint32_t st;         // inflate status of the header
  uint32_t kind;       // kBlkNone: the half sits this round out, kBlkEnd: no header bits left, kBlkStored, kBlkCoded
  uint32_t final;      // BFINAL
  uint32_t len;        // stored: bytes
  uint32_t data_byte;  // stored: the first data byte, counted from the segment's first byte
  uint32_t hdr_end;    // coded: bit offset of the first token code
};
constexpr uint32_t kBlkNone = 0, kBlkEnd = 1, kBlkStored = 2, kBlkCoded = 3;
// Blocks WITH OUTPUT of a segment this kernel follows (the empty stored blocks of a flush are not counted).  Every coded block
// costs the wave its tables and three passes, so a segment cut into dozens of small blocks (zlib with memLevel 1: a block
// every 127 symbols) is the lane-serial kernel's, and is handed over after four turns rather than after dozens.
constexpr uint32_t kSpecMaxBlocks = 4;

template <class L>
__device__ BlockOpen open_block(const uint8_t* src, uint64_t src_n, uint64_t lo, uint64_t hi, uint32_t at, uint8_t* m) {
  using namespace inflate;
  BlockOpen b{kOk, kBlkEnd, 0, 0, 0, 0};
  BitReader br;
  br.open(src, src_n, lo, hi);
  if (at + 3 > br.nbits) return b;  // out of input (inflate::decode_segment's loop head)
  br.seek(at >> 3);
  br.bitpos = at & ~7u;
  br.refill();
  br.drop(at & 7u);
  br.refill();
  b.final = br.get(1);
  const uint32_t type = br.get(2);
  if (type == 3) {
    b.st = kInvalidBlockHeader;
    return b;
  }
  if (type == 0) {  // src/decompress.cpp:416-436
    br.drop((8u - (br.bitpos & 7u)) & 7u);
    br.refill();
    if (br.bitpos + 32 > br.nbits) {
      b.st = kSrcTooSmall;
      return b;
    }
    const uint32_t len = br.get(16);
    br.refill();
    const uint32_t nlen = br.get(16);
    if ((len ^ nlen) != 0xFFFFu) {
      b.st = kNoCompressionLenMismatch;
      return b;
    }
    if (br.bitpos + 8 * len > br.nbits) {
      b.st = kSrcTooSmall;
      return b;
    }
    b.kind = kBlkStored;
    b.len = len;
    b.data_byte = br.bitpos >> 3;
    return b;
  }
  b.st = read_lengths<L>(br, m, type);
  b.kind = kBlkCoded;
  b.hdr_end = br.bitpos;
  return b;
}

// One wave, two segments (32 lanes each), every segment block after block as the serial decoder reads them
// (inflate::decode_segment): a stored block becomes the segment's raw copy (the first block, holding all its bytes) or
// literal tokens; a coded block is decoded by the 32 lanes as described above.  A segment is finished HERE only when
// everything about it was in order; else it is left to k_inflate_tokens (kRetrySerial).
// where a segment stands between its blocks.  In LDS, and re-read after every pass (kept in registers across the passes it
// costs the token loops their registers: 236 bytes of scratch, some of it inside them, 6 % of the kernel's time)
struct SegState {
  uint64_t lo, hi;     // the segment's stream bytes
  uint64_t raw_off;
  uint32_t seg_bits, out_n;
  uint32_t at;         // bit offset of the next block header
  uint32_t out_base, tok_base;  // bytes / tokens of the blocks before
  uint32_t state, raw;
};
enum : uint32_t { kSegRun = 0, kSegDone = 1, kSegRetry = 2 };

__device__ __forceinline__ void tokens_wave_spec(const uint8_t* __restrict__ src, uint64_t src_n, const uint64_t* __restrict__ index,
                                                 uint32_t nseg, uint64_t dst_n, uint32_t* __restrict__ tokens,
                                                 SegInfo* __restrict__ info, uint32_t sps) {
  using L = inflate::SharedLayout;
  __shared__ __align__(16) uint8_t s_tab[2][L::kBytes];
  __shared__ __align__(16) RegionLds s_reg;
  __shared__ BlockOpen s_blk[2];
  __shared__ SegState s_st[2];
  const uint32_t lane = threadIdx.x, half = lane >> 5, hl = lane & 31;
  const uint32_t seg = 2 * blockIdx.x + half;
  {
    // RFC 1951 3.2.5 as a table: base | extra bits << 16
    uint32_t base = 0, extra = 0;
    if (lane < 29) inflate::length_info(257 + lane, base, extra);
    else if (lane >= 32 && lane < 62) inflate::distance_info(lane - 32, base, extra);
    s_reg.lut[lane] = base | (extra << 16);
  }
  if (hl == 0) {
    const bool live = seg < nseg;
    const uint64_t lo = live ? index[seg] : 0, hi = live ? index[seg + 1] : 0;
    const uint64_t obase = (uint64_t)seg * kChunk;
    const uint32_t out_n = (live && dst_n > obase) ? (uint32_t)(dst_n - obase < kChunk ? dst_n - obase : kChunk) : 0u;
    const uint32_t seg_bits = (hi > lo && hi - lo < (1ull << 16)) ? 8u * (uint32_t)(hi - lo) : 0u;
    SegState z;
    z.lo = lo;
    z.hi = hi;
    z.raw_off = 0;
    z.seg_bits = seg_bits;
    z.out_n = out_n;
    z.at = 0;
    z.out_base = 0;
    z.tok_base = 0;
    // not for this kernel: no output, no input, an index that does not hold, 64 KiB or more of stream for 32 KiB of output (a
    // stored segment is 32 KiB + 5 bytes, a coded one smaller: nothing well-formed is that long, and the counting rounds
    // have no business walking what the serial decoder gives up on after 32 KiB of output)
    z.state = !live ? (uint32_t)kSegDone : ((out_n == 0 || seg_bits == 0 || hi > src_n) ? (uint32_t)kSegRetry : (uint32_t)kSegRun);
    z.raw = 0;
    s_st[half] = z;
  }
  const uint32_t hshift = 32 * half;
  SegState& S = s_st[half];
  // (what the compiler knows of S is void behind this: the state is re-read from LDS, not carried in registers)
  auto fresh = []() { asm volatile("" ::: "memory"); };
  // Behind a block: the empty stored blocks a flush leaves (Z_SYNC_FLUSH / Z_FULL_FLUSH, this library's byte alignment) are
  // read on the spot, as the serial decoder would read them, so that the usual segment -- one coded block, one empty stored
  // block -- is one turn of the loop below.  Anything else stays where it is for open_block.  Returns the state; at: in / out.
  auto behind_block = [&](uint32_t& at, bool fin, uint32_t out_base) -> uint32_t {
    const uint8_t* sp = src + S.lo;
    const uint32_t seg_bits = S.seg_bits;
    while (!fin) {
      if (at + 3 > seg_bits) break;
      const uint32_t by = at >> 3;
      const uint32_t h = (uint32_t)sp[by] | (8 * (by + 1) < seg_bits ? (uint32_t)sp[by + 1] << 8 : 0u);
      const uint32_t b3 = (h >> (at & 7u)) & 7u;
      if (b3 >> 1) break;  // a coded block (or an invalid type)
      const uint32_t p = (at + 3u + 7u) & ~7u;
      if (p + 32 > seg_bits) break;
      const uint8_t* q = sp + (p >> 3);
      if (q[0] | q[1] | (uint8_t)~q[2] | (uint8_t)~q[3]) break;  // LEN != 0 or NLEN != ~LEN
      at = p + 32;
      fin = (b3 & 1u) != 0;
    }
    // BFINAL seen, or no header bits left: the segment is complete if it has all its bytes (else the serial decoder says what is wrong)
    if (fin || at + 3 > seg_bits) return out_base == S.out_n ? (uint32_t)kSegDone : (uint32_t)kSegRetry;
    return kSegRun;
  };
#pragma nounroll
  for (uint32_t blk = 0; blk < kSpecMaxBlocks; ++blk) {
    __syncthreads();  // (the tables and records of the block before are read, the states are written)
    fresh();
    if (__ballot(S.state == kSegRun) == 0) break;
    if (hl == 0) {
      if (S.state == kSegRun) s_blk[half] = open_block<L>(src, src_n, S.lo, S.hi, S.at, s_tab[half]);
      else s_blk[half].kind = kBlkNone;
    }
    __syncthreads();
    fresh();
    bool coded = false;
    if (S.state == kSegRun) {
      // every lane of the segment works out the same new state; lane 0 of it writes it down
      const BlockOpen b = s_blk[half];
      uint32_t state = kSegRun, at = S.at, out_base = S.out_base, tok_base = S.tok_base, raw = S.raw;
      uint64_t raw_off = S.raw_off;
      const uint32_t out_n = S.out_n;
      if (b.st != inflate::kOk) state = kSegRetry;
      else if (b.kind == kBlkEnd) state = out_base == out_n ? (uint32_t)kSegDone : (uint32_t)kSegRetry;
      else if (b.kind == kBlkStored) {
        if (b.len > out_n - out_base) state = kSegRetry;
        else {
          // the whole segment is this block (nothing emitted yet -- empty blocks may precede it, as in decode_segment,
          // sf_inflate_core.h): the byte-copy kernel takes it from the stream
          if (out_base == 0 && tok_base == 0 && b.len == out_n && b.len != 0) {
            raw = 1;
            raw_off = S.lo + b.data_byte;
          } else if (raw) {
            state = b.len ? (uint32_t)kSegRetry : state;  // (cannot happen: a raw segment has no bytes left)
          } else {
            uint32_t* seg_tokens = tokens + (uint64_t)seg * kChunk + tok_base;
            const uint8_t* from = src + S.lo + b.data_byte;
            for (uint32_t k = hl; k < b.len; k += 32) seg_tokens[k] = from[k];
            tok_base += b.len;
          }
          out_base += b.len;
          at = 8u * (b.data_byte + b.len);
          if (state == kSegRun) state = behind_block(at, b.final != 0, out_base);
        }
      } else {
        coded = true;
        if (raw || b.hdr_end > S.seg_bits) state = kSegRetry;  // (tokens behind a raw copy: the serial decoder's business)
      }
      coded = coded && state == kSegRun;
      if (hl == 0) {
        S.state = state;
        S.at = at;
        S.out_base = out_base;
        S.tok_base = tok_base;
        S.raw = raw;
        S.raw_off = raw_off;
      }
    }
    // the code tables, by the whole wave for either half (uniform conditions: they sit in LDS)
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
      if (s_blk[h].kind == kBlkCoded && s_blk[h].st == inflate::kOk) {
        build_tables_wave<true>(s_tab[h], lane);
        build_tables_wave<false>(s_tab[h], lane);
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
      if (s_blk[h].kind == kBlkCoded && s_blk[h].st == inflate::kOk) {
        build_limits<true>(s_tab[h], s_reg.lim[h][0], s_reg.base[h][0], lane);
        build_limits<false>(s_tab[h], s_reg.lim[h][1], s_reg.base[h][1], lane);
      }
    }
    __syncthreads();
    fresh();
    const bool take = coded;
    if (__ballot(take) == 0) continue;
    const uint8_t* m = s_tab[half];
    uint32_t entry, nom1;
    RegionOut o;
    // 1. look-back
    {
      const uint32_t hdr = s_blk[half].hdr_end;
      const uint32_t body = take ? S.seg_bits - hdr : 0u;
      const uint32_t nom0 = hdr + (uint32_t)(((uint64_t)body * hl) >> 5);
      nom1 = hdr + (uint32_t)(((uint64_t)body * (hl + 1)) >> 5);
      entry = nom0;
      const uint32_t back = nom0 - hdr < kSpecLookBack ? nom0 - hdr : kSpecLookBack;
      const bool look = take && back != 0;
      decode_regions_lockstep<L, true>(src, src_n, S.lo, S.hi, nom0 - back, nom0, false, 0u, kChunk, nullptr, m, half, 0x40000000u,
                                       look, lane, s_reg, o);
      if (look && o.st == inflate::kOk && !o.eob) entry = o.bit;
    }
    // 2. count, until every lane starts where its predecessor ended.  A lane that has to start over has usually been right
    // from a few hundred bits into its span on (the wrong start fell in with the true chain there), so the first attempt
    // leaves a checkpoint kSpecCheck bits into the span -- a token boundary of ITS chain with the tokens and bytes before it
    // -- and the second attempt stops at the first boundary at or behind it: the same bit, and what the first attempt counted
    // from there on stands (zlib-made streams: four waves in ten start a lane over; a whole pass each before this, a quarter).
    uint32_t exitb = 0, cnt = 0, nbytes = 0, lst = inflate::kOk;
    uint32_t cp_bit = 0, cp_tok = 0, cp_bytes = 0;
    bool leob = false, need = take, settled = false, tried_short = false;
#pragma nounroll
    for (uint32_t round = 0; round < 66; ++round) {
      fresh();
      const bool shortcut = need && cp_bit != 0 && !tried_short && cp_bit > entry;
      // (the bytes the segment has left, at most: no lane can produce more, and a speculative one on a wrong chain stops there)
      decode_regions_lockstep<L, true>(src, src_n, S.lo, S.hi, entry, shortcut ? cp_bit : (nom1 < entry ? entry : nom1),
                                       hl == 31 && !shortcut, 0u, S.out_n - S.out_base, nullptr, m, half, 0x40000000u, need, lane,
                                       s_reg, o, shortcut ? ~0u : entry + kSpecCheck);
      bool again = false;  // the shortcut did not meet the checkpoint: the whole span next round
      if (need && shortcut) {
        tried_short = true;
        if (o.st == inflate::kOk && !o.eob && o.bit == cp_bit) {
          cnt = o.ntok + (cnt - cp_tok);
          nbytes = o.bytes + (nbytes - cp_bytes);
          cp_tok = o.ntok;  // the checkpoint, seen from the new start
          cp_bytes = o.bytes;
        } else {
          again = true;
        }
      } else if (need) {
        exitb = o.bit;
        cnt = o.ntok;
        nbytes = o.bytes;
        lst = o.st;
        leob = o.eob;
        cp_bit = o.cp_bit;
        cp_tok = o.cp_tok;
        cp_bytes = o.cp_bytes;
        tried_short = false;
      }
      const uint32_t pe = (uint32_t)__shfl_up((int)exitb, 1);
      need = take && hl != 0 && (entry != pe || again);
      if (__ballot(need) == 0) {
        settled = true;
        break;
      }
      if (need) {
        tried_short = tried_short && entry == pe;  // (a new start may take the shortcut again)
        entry = pe;
      }
    }
    // every lane is exact now: an error anywhere, a last lane without its end-of-block code, or more bytes than the segment
    // has left leaves the segment to the serial kernel
    uint32_t tsum = cnt, bsum = nbytes;
#pragma unroll
    for (uint32_t d = 1; d < 32; d <<= 1) {
      const uint32_t tu = (uint32_t)__shfl_up((int)tsum, d, 32), bu = (uint32_t)__shfl_up((int)bsum, d, 32);
      if (hl >= d) {
        tsum += tu;
        bsum += bu;
      }
    }
    fresh();
    const uint32_t total_bytes = (uint32_t)__shfl((int)bsum, 31, 32);
    const bool lane_bad = take && (lst != inflate::kOk || (hl == 31 && !leob));
    const uint32_t bad_half = (uint32_t)((__ballot(lane_bad) >> hshift) & 0xFFFFFFFFull);
    const bool go = take && settled && bad_half == 0 && total_bytes <= S.out_n - S.out_base;
    // 3. the spans as a sub-index: the lockstep pass with every check, writing
    {
      const uint32_t ob = S.out_base + bsum - nbytes;
      decode_regions_lockstep<L, false>(src, src_n, S.lo, S.hi, entry, exitb, hl == 31, ob, ob + nbytes,
                                        tokens + (uint64_t)seg * kChunk + S.tok_base + (tsum - cnt), m, half, (seg % sps) * kChunk, go,
                                        lane, s_reg, o);
    }
    fresh();
    const bool lane_failed = go && (o.st != inflate::kOk || o.ntok != cnt);
    const uint32_t failed_half = (uint32_t)((__ballot(lane_failed) >> hshift) & 0xFFFFFFFFull);
    uint32_t at = (uint32_t)__shfl((int)o.bit, 31, 32);  // behind the end-of-block code
    const uint32_t total_tok = (uint32_t)__shfl((int)tsum, 31, 32);
    if (take) {
      uint32_t state = kSegRetry, out_base = S.out_base, tok_base = S.tok_base;
      if (go && !failed_half) {
        out_base += total_bytes;
        tok_base += total_tok;
        state = behind_block(at, s_blk[half].final != 0, out_base);
      }
      if (hl == 0) {
        S.state = state;
        S.at = at;
        S.out_base = out_base;
        S.tok_base = tok_base;
      }
    }
  }
  __syncthreads();
  fresh();
  if (hl == 0 && seg < nseg) {
    SegInfo si;
    si.status = S.state == kSegDone ? (uint32_t)inflate::kOk : kRetrySerial;  // (still running: more blocks than this kernel follows)
    si.ntok = S.raw ? 0u : S.tok_base;
    si.raw = S.raw;
    si.out_n = S.out_n;
    si.raw_off = S.raw_off;
    info[seg] = si;
  }
}

__attribute__((amdgpu_waves_per_eu(5, 5)))
__global__ __launch_bounds__(64, 2) void k_inflate_tokens_spec(const uint8_t* __restrict__ src, uint64_t src_n,
                                                              const uint64_t* __restrict__ index, uint32_t nseg,
                                                              uint64_t dst_n, uint32_t* __restrict__ tokens,
                                                              SegInfo* __restrict__ info, uint32_t sps) {
  tokens_wave_spec(src, src_n, index, nseg, dst_n, tokens, info, sps);
}

// inclusive wave scan on the DPP network (row_shr 1/2/4/8, row_bcast 15/31): no LDS round trips
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_from(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t /*lane*/) {
  v = dpp_add_from<0x111, 0xF>(v);
  v = dpp_add_from<0x112, 0xF>(v);
  v = dpp_add_from<0x114, 0xF>(v);
  v = dpp_add_from<0x118, 0xF>(v);
  v = dpp_add_from<0x142, 0xA>(v);
  v = dpp_add_from<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v, 0), 63);
}

__device__ __forceinline__ uint32_t load_word_guarded(const uint8_t* base, uint64_t src_n, uint64_t w) {
  const uint64_t b = 4 * w;
  if (b + 4 <= src_n) return reinterpret_cast<const uint32_t*>(base)[w];
  uint32_t v = 0;
  for (uint32_t k = 0; k < 4; ++k)
    if (b + k < src_n) v |= (uint32_t)base[b + k] << (8 * k);
  return v;
}

// The byte-copy half (src/decompress.cpp:157-187,388-398).  Tokens are placed in steps of up to 1536 tokens (three per thread) /
// 3968 output bytes by a workgroup prefix sum; a bitmap of token starts plus per-word popcount prefixes then
// lets every THREAD TAKE BYTES, not tokens: byte j finds its token with two broadcast reads and a popcount, so
// all lanes work whatever the match lengths are.  A match byte whose source lies before the step is final
// already and is copied at once; one whose source lies inside the step gets a 16-bit pointer to it, and pointer
// jumping (ptr[j] = ptr[ptr[j]], barrier-separated rounds, at most log2(3968)) takes every such byte to a final
// one, however the matches of the step nest or overlap themselves; then the byte is fetched.  51 KiB of LDS (the
// 32 KiB window + one step of pointers and token records): three workgroups share a CU and hide each other's
// barriers.
// One segment.  The output window is a ring of KB_RING bytes in LDS: the 32 KiB a match may reach back plus the
// step being produced; byte p of the strip lives at p mod KB_RING.  Every step's bytes go to `dst` as soon as they
// are final (whole dwords; the odd bytes with the next step), so the ring never has to hold a whole segment.
// rb: ring position of the segment's first byte; segbase: that byte's position in its strip (bytes of history).
// Returns false when the segment failed.
__device__ bool inflate_segment_bytes(const uint8_t* __restrict__ src, uint64_t src_n, const uint32_t* __restrict__ tokens,
                                      SegInfo* __restrict__ info, uint8_t* __restrict__ dst, uint32_t seg, uint32_t segbase,
                                      uint32_t rb, uint8_t* s_dyn) {
  constexpr uint32_t kRing = KB_RING;
  uint8_t* s_out = s_dyn;                                          // [kRing] output window
  uint16_t* s_ptr = reinterpret_cast<uint16_t*>(s_dyn + kRing);   // [KB_SPAN] step-relative source, or kFinal
  uint32_t* s_tinfo = reinterpret_cast<uint32_t*>(s_dyn + kRing + 2 * KB_SPAN);  // [1024] start | match | byte or dist-1
  uint32_t* s_mark = s_tinfo + KB_THREADS * KB_TPT;                // [KB_SPAN / 32] bit: a token starts here
  uint32_t* s_wpre = s_mark + KB_SPAN / 32;                        // [KB_SPAN / 32] tokens starting before the word
  uint32_t* s_w = s_wpre + KB_SPAN / 32;                           // [8] wave totals, [2] next step
  uint32_t* s_next = s_w + KB_THREADS / 64;
  uint32_t* s_any = s_next + 2;                                    // [4] wg_any's flags
  static_assert(4 * (KB_THREADS / 64 + 2 + 4) <= 64, "the tail of KB_AUX holds the wave totals, the step's end and the flags");
  constexpr uint32_t kFinal = 0xFFFFu;
  constexpr uint32_t kWords = KB_SPAN / 32;
  constexpr uint32_t kListCap = KB_THREADS * KB_TPT * 4 / 2 / (KB_THREADS / 64);  // pointer bytes a wave can list (384)
  static_assert(kWords <= 128 && KB_SPAN % 32 == 0, "one wave scans the bitmap, two words per lane");
  const uint32_t t = threadIdx.x, lane = t & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // ring index of the segment's byte q (q < kChunk + kRing)
  auto ring = [&](uint32_t q) -> uint32_t {
    uint32_t a = rb + q;
    a = a >= kRing ? a - kRing : a;
    return a >= kRing ? a - kRing : a;
  };
  // a < 2 * kRing -> a mod kRing in two instructions: the unsigned minimum of a and a - kRing (which wraps to a huge
  // number below kRing); likewise a - d for d <= a + kRing
  auto wrap1 = [](uint32_t a) -> uint32_t { return min(a, a - kRing); };
  auto back1 = [](uint32_t a, uint32_t d) -> uint32_t { return min(a - d, a - d + kRing); };  // a < kRing, d <= kWindow + a
  const SegInfo si = info[seg];
  if (si.status != inflate::kOk) return false;
  const uint32_t out_n = si.out_n;
  uint8_t* o = dst + (uint64_t)seg * kChunk;  // 16-byte aligned

  if (si.raw & kSegRaw) {
    // stored segment: dword copy from an arbitrarily aligned stream position
    const uint32_t mis = (uint32_t)(si.raw_off & 3);
    const uint64_t w0 = si.raw_off >> 2;
    const uint32_t nd = out_n / 4;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
    for (uint32_t k = t; k < nd; k += KB_THREADS) {
      const uint32_t lo = load_word_guarded(src, src_n, w0 + k);
      const uint32_t hi = mis ? load_word_guarded(src, src_n, w0 + k + 1) : 0u;
      o32[k] = __builtin_amdgcn_alignbyte(hi, lo, mis);
    }
    const uint32_t done = 4 * nd;
    if (t < out_n - done) o[done + t] = src[si.raw_off + done + t];
    {
      // later segments of the strip may copy from these bytes: their last kWindow go into the ring as well
      __syncthreads();  // (the previous segment's last window reads are done)
      const uint32_t k0 = out_n > kWindow ? (out_n - kWindow) / 4 : 0u;  // (rb and 4 * k are multiples of 4: dwords do not wrap)
      for (uint32_t k = k0 + t; k < nd; k += KB_THREADS) {
        const uint32_t lo = load_word_guarded(src, src_n, w0 + k);
        const uint32_t hi = mis ? load_word_guarded(src, src_n, w0 + k + 1) : 0u;
        *reinterpret_cast<uint32_t*>(s_out + ring(4 * k)) = __builtin_amdgcn_alignbyte(hi, lo, mis);
      }
      if (t < out_n - done) s_out[ring(done + t)] = src[si.raw_off + done + t];
      __syncthreads();
    }
    return true;
  }

  const uint32_t ntok = si.ntok;
  const uint32_t* tk = tokens + (uint64_t)seg * kChunk;
  uint32_t tok_base = 0, pos0 = 0;  // uniform: first token / output byte of the step
  uint32_t flushed = 0;             // uniform: bytes of the segment already in dst (a multiple of 4)
  bool bad = false;
  if (t == 0) {
    s_next[0] = 0;  // tokens placed by the step
    s_next[1] = 0;  // where its output ends
  }
  // "does any thread of the workgroup say so", with ONE barrier (__syncthreads_or is three: the library clears its word,
  // ORs into it and reads it, a barrier after each; at two to three of them per step that was half the kernel's barriers).
  // Four flag words in rotation: a call stores 1 into its word where a wave has a taker, and clears the word of two calls
  // on -- last read before the previous call's barrier, next written behind this one's.  The barrier is the phase barrier
  // the call sites need anyway.
  if (t < 4) s_any[t] = 0;  // (the caller's barrier between segments / the first step's barrier orders this)
  uint32_t any_gen = 0;     // uniform
  auto wg_any = [&](bool pred) -> bool {
    const uint32_t slot = any_gen & 3u;
    ++any_gen;
    if (__ballot(pred) != 0 && lane == 0) s_any[slot] = 1u;
    if (t == 64) s_any[(slot + 2u) & 3u] = 0u;
    __syncthreads();
    return s_any[slot] != 0u;
  };
  if (t < kWords) s_mark[t] = 0;
  uint32_t next[KB_TPT];
#pragma unroll
  for (uint32_t k = 0; k < KB_TPT; ++k) next[k] = KB_TPT * t + k < ntok ? tk[KB_TPT * t + k] : 0u;
  while (tok_base < ntok) {
    // ---- place up to KB_TPT tokens per thread (loaded while the previous step was being resolved) ----
    const uint32_t i0 = tok_base + KB_TPT * t;
    uint32_t tok[KB_TPT], len[KB_TPT], start[KB_TPT];
    bool val[KB_TPT], mat[KB_TPT], fit[KB_TPT];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < KB_TPT; ++k) {
      tok[k] = next[k];
      val[k] = i0 + k < ntok;
      mat[k] = val[k] && (tok[k] >> 31);
      len[k] = val[k] ? (mat[k] ? ((tok[k] >> 16) & 0xFFu) + 3u : 1u) : 0u;
      mine += len[k];
    }
    const uint32_t incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
#pragma unroll
    for (uint32_t w = 0; w < KB_THREADS / 64; ++w)
      if (w < wave) pre += s_w[w];
    const uint32_t limit = pos0 + KB_SPAN < out_n ? pos0 + KB_SPAN : out_n;
    // the tokens that end inside the span form a prefix of the step (the first always fits: 258 <= KB_SPAN)
    // k_inflate_tokens has validated every token; the checks keep a corrupted token buffer inside the window
    uint32_t nfit = 0, fmax = 0;
    {
      uint32_t at = pos0 + pre + incl - mine;
#pragma unroll
      for (uint32_t k = 0; k < KB_TPT; ++k) {
        start[k] = at;
        at += len[k];
        fit[k] = val[k] && at <= limit;
        const uint32_t dist = (tok[k] & 0x7FFFu) + 1u;
        if ((fit[k] && mat[k] && dist > start[k] + segbase) || (val[k] && at > out_n)) bad = true;
        if (fit[k]) { fmax = at; ++nfit; }
      }
    }
    {
      // where the next step starts: the fitting tokens are a prefix, so their count and the largest end say it
      const uint32_t nfit_w = wave_sum_u32(nfit);
      const uint64_t fitting = __ballot(nfit != 0);
      if (fitting) {
        // (a prefix of the tokens fits: the wave's last fitting lane knows the largest end)
        const uint32_t fmax_w = (uint32_t)__builtin_amdgcn_readlane((int)fmax, 63 - __builtin_clzll(fitting));
        if (lane == 0) {
          atomicAdd(&s_next[0], nfit_w);
          atomicMax(&s_next[1], fmax_w);
        }
      }
    }
    // token records and the bitmap of token starts (step-relative)
#pragma unroll
    for (uint32_t k = 0; k < KB_TPT; ++k) {
      if (fit[k] && !bad) {
        const uint32_t r = start[k] - pos0;
        s_tinfo[KB_TPT * t + k] = r | (mat[k] ? 0x1000u | ((tok[k] & 0x7FFFu) << 16) : (tok[k] & 0xFFu) << 16);
        atomicOr(&s_mark[r >> 5], 1u << (r & 31));
      }
    }
    if (wg_any(bad)) {
      if (t == 0) info[seg].status = inflate::kError;
      return false;
    }
    const uint32_t next_tok = tok_base + s_next[0], next_pos = s_next[1];
    const uint32_t span_n = next_pos - pos0;
    {
      const uint32_t n0 = next_tok + KB_TPT * t;  // the next step's tokens: in flight during paint and jumping
#pragma unroll
      for (uint32_t k = 0; k < KB_TPT; ++k) next[k] = n0 + k < ntok ? tk[n0 + k] : 0u;
    }
    if (wave == 0) {
      // tokens starting before each bitmap word: one wave, two words per lane
      const uint32_t a = 2 * lane < kWords ? (uint32_t)__popc(s_mark[2 * lane]) : 0u;
      const uint32_t b2 = 2 * lane + 1 < kWords ? (uint32_t)__popc(s_mark[2 * lane + 1]) : 0u;
      const uint32_t inc2 = wave_scan_incl(a + b2, lane);
      if (2 * lane < kWords) s_wpre[2 * lane] = inc2 - a - b2;
      if (2 * lane + 1 < kWords) s_wpre[2 * lane + 1] = inc2 - b2;
    }
    __syncthreads();
    // ---- every thread takes bytes: find the token, then literal / final copy / step-relative pointer ----
    const uint32_t pbase = ring(pos0);  // (uniform) ring position of the step's first byte: byte j lives at wrap1(pbase + j)
    uint32_t pm = 0;  // which of the thread's bytes got a pointer
#pragma unroll
    for (uint32_t i = 0; i < (KB_SPAN + KB_THREADS - 1) / KB_THREADS; ++i) {
      const uint32_t j = t + KB_THREADS * i;
      if (j < span_n) {
        const uint32_t w = j >> 5;
        const uint32_t idx = s_wpre[w] + (uint32_t)__popc(s_mark[w] & (0xFFFFFFFFu >> (31u - (j & 31u)))) - 1u;
        const uint32_t ti = s_tinfo[idx];
        const uint32_t cur = wrap1(pbase + j);
        // No branch on what the byte is (a wave executes every side of such a branch anyway): a literal is final, a match
        // byte whose source lies before the step is final and copied at once (dist <= kWindow < kRing), one whose source
        // lies inside the step gets a pointer -- and a byte that nobody reads before the resolve pass overwrites it (what
        // a literal reads, one byte back, it ignores)
        const bool mt = (ti & 0x1000u) != 0;
        const uint32_t dist = (ti >> 16) + 1u;
        const uint32_t sv = s_out[back1(cur, mt ? dist : 1u)];
        s_out[cur] = (uint8_t)(mt ? sv : ti >> 16);
        const bool fin = !mt || dist > j;
        s_ptr[j] = (uint16_t)(fin ? kFinal : j - dist);
        pm |= (fin ? 0u : 1u) << i;
      }
    }
    // The bytes with pointers are a third of a step of text: each wave lists its own (kListCap entries in what were the
    // token records, dead now; the list is the wave's alone, so no barrier stands between writing and reading it) and
    // jumping and resolving run over the lists, all lanes busy, instead of over every byte (round 5: 1.25 + 0.3 ms of the
    // kernel's 3.25 per GiB went into those two passes).  A step with more pointers than the lists hold -- runs,
    // short periods -- takes the passes over every byte, as before.
    const uint32_t pcnt = (uint32_t)__popc(pm);
    const uint32_t pincl = wave_scan_incl(pcnt, lane);
    const uint32_t wtotal = (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);
    if (!wg_any(wtotal > kListCap)) {
      uint16_t* lst = reinterpret_cast<uint16_t*>(s_tinfo) + wave * kListCap;
      {
        uint32_t at = pincl - pcnt;
#pragma unroll
        for (uint32_t i = 0; i < (KB_SPAN + KB_THREADS - 1) / KB_THREADS; ++i)
          if ((pm >> i) & 1u) lst[at++] = (uint16_t)(t + KB_THREADS * i);
      }
      // The kernel is bound by its LDS instructions (scattered 16-bit accesses), so a lane stops hopping as soon as it
      // reads "final" (two thirds of the entries at their first probe) and an entry that has reached a final byte sits the
      // later rounds out (1.4 rounds per step of text); up to four hops a round (2 / 3 / 4 / 6 unconditional hops on one
      // box: 2.78 / 2.75 / 2.68 / 2.72 ms per GiB)
      uint32_t done = 0;  // bit i: the lane's i-th list entry (kListCap / 64 = 6 at most)
      for (;;) {  // ends: every change moves a pointer to a strictly smaller index; afterwards every pointer is at a final byte
        bool changed = false;
        uint32_t it = 0;
        for (uint32_t k = lane; k < wtotal; k += 64, ++it) {
          if (!((done >> it) & 1u)) {
            const uint32_t j = lst[k];
            const uint32_t p = s_ptr[j];
            uint32_t cur = p, nxt = s_ptr[cur];
#pragma unroll
            for (uint32_t h = 1; h < 4; ++h) {
              if (nxt != kFinal) {
                cur = nxt;
                nxt = s_ptr[cur];
              }
            }
            if (cur != p) s_ptr[j] = (uint16_t)cur;
            if (nxt == kFinal) done |= 1u << it;
            else changed = true;  // what the last hop reached is a pointer itself
          }
        }
        if (!wg_any(changed)) break;
      }
      for (uint32_t k = lane; k < wtotal; k += 64) {
        const uint32_t j = lst[k];
        s_out[wrap1(pbase + j)] = s_out[wrap1(pbase + s_ptr[j])];
      }
    } else {
    // ---- pointer jumping inside the step; a pointer only ever moves to an ancestor, so in place is fine ----
    for (;;) {  // ends: every change moves a pointer to a strictly smaller index; afterwards every entry is final or points at one
      bool changed = false;
#pragma unroll
      for (uint32_t i = 0; i < (KB_SPAN + KB_THREADS - 1) / KB_THREADS; ++i) {
        const uint32_t j = t + KB_THREADS * i;
        if (j < span_n) {
          // two hops per round (a pointer only ever moves to an ancestor, whatever the other threads have done to it
          // meanwhile), without branches: a final entry reads itself and is written back as it was
          const uint32_t p = s_ptr[j];
          const uint32_t q = s_ptr[p != kFinal ? p : j];
          const uint32_t r2 = s_ptr[q != kFinal ? q : j];
          const bool upd = p != kFinal && q != kFinal;
          s_ptr[j] = (uint16_t)(upd ? (r2 != kFinal ? r2 : q) : p);
          // another round only for an entry that still points at a pointer: one that has just reached a final byte
          // (r2 final: it points at q now, and a final entry stays final) is done, so the round that finishes the
          // last entry also ends the loop (round 5; before, one more round in which nothing changed)
          changed = changed || (upd && r2 != kFinal);
        }
      }
      if (!wg_any(changed)) break;
    }
#pragma unroll
    for (uint32_t i = 0; i < (KB_SPAN + KB_THREADS - 1) / KB_THREADS; ++i) {
      const uint32_t j = t + KB_THREADS * i;
      if (j < span_n) {
        const uint32_t p = s_ptr[j];  // (final since the paint phase, or a pointer to a byte that is)
        const uint32_t cur = wrap1(pbase + j);
        s_out[cur] = s_out[p != kFinal ? wrap1(pbase + p) : cur];  // (a final byte is written back as it is)
      }
    }
    }
    if (t == 0) {
      s_next[0] = 0;
      s_next[1] = 0;
    }
    if (t < kWords) s_mark[t] = 0;
    __syncthreads();
    // the step's bytes are final: whole dwords go out now (ring and dst are dword-aligned alike)
    {
      const uint32_t end4 = next_pos & ~3u;
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
      const uint32_t fbase = ring(flushed);  // (uniform; the bytes to flush span less than kRing)
      for (uint32_t q = flushed + 4 * t; q < end4; q += 4 * KB_THREADS)
        o32[q >> 2] = *reinterpret_cast<const uint32_t*>(s_out + wrap1(fbase + (q - flushed)));
      flushed = end4;
    }
    tok_base = next_tok;
    pos0 = next_pos;
  }
  if (pos0 != out_n) {
    if (t == 0) info[seg].status = inflate::kError;
    return false;
  }
  if (t < out_n - flushed) o[flushed + t] = s_out[ring(flushed + t)];  // the last odd bytes
  return true;
}

// The byte-copy kernel: one workgroup per strip of `sps` segments (sps = 1: independent segments), the
// segments in order, the window carried in LDS from one to the next.
__global__ __launch_bounds__(KB_THREADS) void k_inflate_bytes(const uint8_t* __restrict__ src, uint64_t src_n,
                                                              const uint32_t* __restrict__ tokens,
                                                              SegInfo* __restrict__ info, uint8_t* __restrict__ dst,
                                                              uint32_t nseg, uint32_t sps) {
  extern __shared__ __align__(16) uint8_t s_dyn[];
  const uint32_t seg0 = blockIdx.x * sps;
  uint32_t rb = 0;  // ring position of the segment's first byte: (k * kChunk) mod KB_RING
  for (uint32_t k = 0; k < sps && seg0 + k < nseg; ++k) {
    if (!inflate_segment_bytes(src, src_n, tokens, info, dst, seg0 + k, k * kChunk, rb, s_dyn)) {
      // the later segments of the strip depend on this one: they fail with it (first failure in stream
      // order is what the caller sees, k_inflate_status)
      for (uint32_t j = k + 1 + threadIdx.x; j < sps && seg0 + j < nseg; j += KB_THREADS)
        if (info[seg0 + j].status == inflate::kOk) info[seg0 + j].status = inflate::kError;
      return;
    }
    rb += kChunk;
    rb = rb >= KB_RING ? rb - KB_RING : rb;
    __syncthreads();  // the segment's window writes precede the next segment's reads
  }
}

constexpr uint32_t KS_THREADS = 1024;
__global__ __launch_bounds__(KS_THREADS) void k_inflate_status(const SegInfo* __restrict__ info, uint32_t nseg,
                                                               uint32_t* __restrict__ result /* [status, segment] */) {
  __shared__ uint32_t s_first[KS_THREADS];
  const uint32_t t = threadIdx.x;
  uint32_t first = 0xFFFFFFFFu;
  for (uint32_t s = t; s < nseg; s += KS_THREADS)
    if (info[s].status != inflate::kOk && s < first) first = s;
  s_first[t] = first;
  __syncthreads();
  for (uint32_t o = KS_THREADS / 2; o; o >>= 1) {
    if (t < o && s_first[t + o] < s_first[t]) s_first[t] = s_first[t + o];
    __syncthreads();
  }
  if (t == 0) {
    const uint32_t f = s_first[0];
    result[0] = f == 0xFFFFFFFFu ? (uint32_t)inflate::kOk : info[f].status;
    result[1] = f;
  }
}

}  // namespace

hipError_t init_inflate_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_tokens),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)KT_LDS);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_bytes), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)KB_LDS);
}

// Index-only streams: the speculative wave kernel, then the lane-serial one for the segments that one left (speculate =
// false: the lane-serial kernel for every segment, the path of rounds 2-4; SFH_INFLATE_SERIAL=1).
hipError_t launch_inflate_tokens(const uint8_t* src, uint64_t src_n, const uint64_t* index, uint32_t nseg, uint64_t dst_n,
                                 uint32_t* tokens, SegInfo* info, uint32_t sps, bool speculate, hipStream_t s) {
  if (speculate) {
    hipLaunchKernelGGL(k_inflate_tokens_spec, dim3((nseg + 1) / 2), dim3(64), 0, s, src, src_n, index, nseg, dst_n, tokens,
                       info, sps);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_inflate_tokens, dim3((nseg + KT_LANES - 1) / KT_LANES), dim3(KT_LANES), KT_LDS, s, src, src_n, index,
                     nseg, dst_n, tokens, info, sps, speculate ? 1u : 0u);
  return hipGetLastError();
}

hipError_t launch_inflate_tokens_sub(const uint8_t* src, uint64_t src_n, const uint64_t* index, const uint32_t* subidx,
                                     uint32_t nseg, uint64_t dst_n, uint32_t* tokens, SegInfo* info, uint32_t sps,
                                     hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_tokens_sub, dim3((nseg + 1) / 2), dim3(64), 0, s, src, src_n, index, subidx, nseg, dst_n,
                     tokens, info, sps);
  return hipGetLastError();
}

hipError_t launch_inflate_bytes(const uint8_t* src, uint64_t src_n, uint32_t nseg, const uint32_t* tokens, SegInfo* info,
                                uint8_t* dst, uint32_t sps, hipStream_t s) {
  const uint32_t nstrips = (nseg + sps - 1) / sps;
  hipLaunchKernelGGL(k_inflate_bytes, dim3(nstrips), dim3(KB_THREADS), KB_LDS, s, src, src_n, tokens, info, dst, nseg, sps);
  return hipGetLastError();
}

hipError_t launch_inflate_status(const SegInfo* info, uint32_t nseg, uint32_t* d_result, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_status, dim3(1), dim3(KS_THREADS), 0, s, info, nseg, d_result);
  return hipGetLastError();
}

}  // namespace sf
