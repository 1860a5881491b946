// sf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DEFLATE encoder.
// Integer/bit work only: no MFMA.  See sf_device.h for the pipeline and DESIGN.md for
// the LDS layout and the per-kernel roofline.  The bitstream these kernels emit is the
// inverse of /root/reference/src/decompress.cpp (contract: SURVEY.md Appendix A).
#include "sf_device.h"

#include <stdlib.h>

namespace sf {

// ---------------------------------------------------------------------------
// symbol arithmetic (inverse of length_infos / distance_infos,
// /root/reference/src/decompress.cpp:53-84)
// ---------------------------------------------------------------------------
// l3 = len-3 in [0,255] -> lit/len symbol, extra-bit count, extra value
__device__ __forceinline__ uint32_t len_symbol(uint32_t l3, uint32_t& ebits, uint32_t& eval) {
  if (l3 == 255) { ebits = 0; eval = 0; return 285; }
  if (l3 < 8) { ebits = 0; eval = 0; return 257 + l3; }
  const uint32_t e = (31 - __builtin_clz(l3)) - 2;
  ebits = e;
  eval = l3 & ((1u << e) - 1);
  return 257 + 4 * e + 4 + ((l3 >> e) & 3);
}
// d1 = dist-1 in [0,32767] -> distance symbol, extra-bit count, extra value
__device__ __forceinline__ uint32_t dist_symbol(uint32_t d1, uint32_t& ebits, uint32_t& eval) {
  if (d1 < 4) { ebits = 0; eval = 0; return d1; }
  const uint32_t hb = 31 - __builtin_clz(d1);
  const uint32_t e = hb - 1;
  ebits = e;
  eval = d1 & ((1u << e) - 1);
  return 2 * hb + ((d1 >> e) & 1);
}
// the symbol alone, without a branch (0 and 1 are their own symbols; from 2 on the general rule holds)
__device__ __forceinline__ uint32_t dist_symbol_of(uint32_t d1) {
  const uint32_t x = d1 > 2u ? d1 : 2u;
  const uint32_t hb = 31 - __builtin_clz(x);
  const uint32_t s = 2 * hb + ((x >> (hb - 1)) & 1);
  return d1 < 2u ? d1 : s;
}
__device__ __forceinline__ uint32_t len_extra_of_sym(uint32_t k /*sym-257*/) {
  return (k < 8 || k == 28) ? 0 : (k - 4) >> 2;
}
__device__ __forceinline__ uint32_t dist_extra_of_sym(uint32_t s) { return s < 4 ? 0 : (s - 2) >> 1; }
__device__ __forceinline__ uint32_t fixed_ll_len(uint32_t s) {
  return s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
}

// Wave-wide scans on the DPP network (no LDS round trips).  Identity 0; a lane whose source does not exist keeps
// the identity (`old` operand, bound_ctrl off).  row_shr:n = 0x110+n, row_bcast:15 = 0x142 (rows 1 and 3),
// row_bcast:31 = 0x143 (rows 2 and 3), wave_shr:1 = 0x138.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_from(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
  v += dpp_from<0x111, 0xF>(v);
  v += dpp_from<0x112, 0xF>(v);
  v += dpp_from<0x114, 0xF>(v);
  v += dpp_from<0x118, 0xF>(v);
  v += dpp_from<0x142, 0xA>(v);
  v += dpp_from<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_excl_max(uint32_t v) {
  v = max(v, dpp_from<0x111, 0xF>(v));
  v = max(v, dpp_from<0x112, 0xF>(v));
  v = max(v, dpp_from<0x114, 0xF>(v));
  v = max(v, dpp_from<0x118, 0xF>(v));
  v = max(v, dpp_from<0x142, 0xA>(v));
  v = max(v, dpp_from<0x143, 0xC>(v));
  return dpp_from<0x138, 0xF>(v);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_add(v), 63);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t /*lane*/) { return wave_incl_add(v); }

// est_log2(x) ~ 256 * log2(x) for x >= 1: the exponent and the top six mantissa bits through a table -- plain integers, the
// same in the specification (oracle: est_log2), so the stored-without-a-code rule of k_plan and the stored-by-the-probe rule of k_lz77 decide alike on both sides
__constant__ uint8_t c_est_lg64[64] = {0, 6, 11, 17, 22, 28, 33, 38, 44, 49, 54, 59, 63, 68, 73, 78, 82, 87, 92, 96, 100, 105, 109, 113, 118, 122,
                                       126, 130, 134, 138, 142, 146, 150, 154, 157, 161, 165, 169, 172, 176, 179, 183, 186, 190, 193, 197,
                                       200, 203, 207, 210, 213, 216, 220, 223, 226, 229, 232, 235, 238, 241, 244, 247, 250, 253};
__device__ __forceinline__ uint32_t est_log2(uint32_t x) {
  const uint32_t e = 31 - __builtin_clz(x | 1u);
  const uint32_t m = (e >= 6 ? x >> (e - 6) : x << (6 - e)) & 63u;
  return (e << 8) + c_est_lg64[m];
}
constexpr uint32_t kEstHeaderBits = 17 + 3 * 4;  // block header, HLIT, HDIST, HCLEN and four code-length-code lengths: the least a dynamic header takes
constexpr uint32_t kStoreMargin = 64;            // bytes: an estimate this close to the stored size settles for "stored"

// ---------------------------------------------------------------------------
// K1: LZ77 match finding + parse + histogram.  One 1024-thread workgroup per STRIP (block_bytes of
// input), two workgroups per CU (LDS <= 80 KiB, 64 VGPRs).  The strip is processed in ROUNDS of
// kRound positions; LDS holds the kWindow bytes before the round, the round itself and a short
// look-ahead, shifted down by one round at every round start, so a position's LDS address is
// kWindow + (position - round start) and a candidate's is that minus the distance.  The hash table
// and the window persist over the strip's DEFLATE blocks (one per kChunk = 8 rounds): matches reach
// into earlier blocks of the strip (legal: /root/reference/src/decompress.cpp:178 only requires
// distance <= bytes written).  The CU has ONE scalar unit for all its waves, so the hot loops are
// straight-line vector code.  Per round:
//   stage : shift the window, store the prefetched 8 KiB, age the table's step codes if due
//   match : 8 steps of step-synchronous hash insertion (kStep positions, one per thread, between
//           two barriers).  A bucket is one dword = two 16-bit history levels {newest, the one
//           before}; ONE ds_max_u32 of (code << 16 | old newest) inserts: every thread of a step
//           carries the same low half, so the result is independent of thread order.  Branch-free
//           ranking by kRank bytes of both far levels (read before the insertion) and the near
//           candidate (the step's first same-hash position, read after it); the winner's next
//           kCap - kRank bytes only where its rank is full.  4-bit capped lengths + 16-bit distances
//   parse : wave-local, in registers: thread t owns the eight positions [8t, 8t + 8) of the round, a
//           wave one kRegion-byte parse region (matches never cross it).  take: the positions whose
//           match the greedy / lazy rule would take; walk: every lane follows the chain through its
//           positions from a speculative entry, the true entry is the exclusive prefix maximum of
//           the exits before it (DPP scan), lanes whose entry changed walk again
//   emit  : the lane's tokens -> compact 16-bit items (literals slot by slot, its at most two matches
//           in two compact rounds) + LDS histogram
// ---------------------------------------------------------------------------
constexpr uint32_t K1_THREADS = kStep;
constexpr uint32_t K1_WAVES = K1_THREADS / 64;
constexpr uint32_t kRound = 8192;                            // positions per match->parse round
constexpr uint32_t kRSubs = kRound / kSubBytes;              // sub-index regions per round
constexpr uint32_t kRoundsPerChunk = kChunk / kRound;
constexpr uint32_t kLook = 32;                               // bytes staged beyond the round (compare + alignment)
constexpr uint32_t kRank = 8;                                // bytes that rank a position's candidates
constexpr uint32_t kLongBytes = 7;                           // bytes the second table (SFH_EFFORT_MAX) is keyed by
constexpr uint32_t kSkipSpan = 8192;                         // stored fast path: decided after this many positions of a chunk
static_assert(kRound / 8 == K1_THREADS && kRegion == 8 * 64, "parse: eight positions per thread, one region per wave");
static_assert(kSubBytes % kRegion == 0 && kRound % kSubBytes == 0, "sub-index regions are whole parse regions");
static_assert(kSkipSpan % kRound == 0 && kChunk % kRound == 0, "round geometry");
// step codes in a 16-bit table half: ((step - epoch) + 1) << 10 | (1023 - t); the epoch advances by
// kEpochSteps whenever a round would reach kEpochMax steps past it, entries older than that vanish
constexpr uint32_t kEpochSteps = 16, kEpochMax = 48;
static_assert(kEpochMax - kEpochSteps >= kWindow / kStep && ((kEpochMax + 1) << 10) <= 65536, "step codes");
static_assert(kEpochMax % (kRound / kStep) == 0 && kEpochSteps % (kRound / kStep) == 0, "ageing happens between rounds");
// LDS carve (bytes); every offset is a multiple of 16
constexpr uint32_t L_DATA = 0;                               // window | round | look-ahead
constexpr uint32_t L_LEN4 = L_DATA + kWindow + kRound + kLook;  // 4 bits per position: capped len-3 (0: no match), eight per dword
constexpr uint32_t L_TABLE = L_LEN4 + kRound / 2 + 16;       // u32[1<<kHashBits]
constexpr uint32_t L_HIST = L_TABLE + (4u << kHashBits);     // u32[320]
constexpr uint32_t L_WTOT = L_HIST + 4 * kHistStride;        // u32[16] tokens | matches << 16 of each wave's region
constexpr uint32_t L_DSYM = L_WTOT + 4 * K1_WAVES;            // u8[512] distance - 1 -> symbol: [d] below 256, [256 + (d >> 7)] from there on
constexpr uint32_t K1_LDS = L_DSYM + 512;
static_assert(2 * K1_LDS <= 160 * 1024, "K1: two workgroups per CU");
// RECENT (SFH_EFFORT_RECENT): exact recency.  A bucket is {hi, lo}: lo the LATEST position with the hash, hi what lo held
// before the most recent step that inserted the hash; positions go in by ordered exchanges (one wave, slice after slice,
// as with the chains), and what an exchange returns -- the position's exact predecessor -- is its near candidate.  The
// hashes the threads POST for that wave (u16 per position) lie where the histogram is: nothing touches the histogram
// during the match phase, its counts wait in registers meanwhile.  The ANSWERS (u16 per searched position) follow the
// distance-symbol table.  Same two workgroups per CU.
constexpr uint32_t R_L_POST = L_HIST;                        // u16[kStep]
constexpr uint32_t R_L_ANS = L_DSYM + 512;                   // u16[kStep / 2]
constexpr uint32_t R_K1_LDS = R_L_ANS + kStep;
static_assert(2 * kStep <= 4 * kHistStride && 2 * R_K1_LDS <= 160 * 1024 && R_L_ANS % 4 == 0, "K1 (recent): two workgroups per CU");
// CHAIN (SFH_EFFORT_BEST / _ULTRA / _EXTREME): exact hash chains.  One workgroup per CU: behind the 4-bit lengths come the
// chain HEADS (one dword per hash holding a 16-bit step code: the LATEST position with it; a dword because they are
// updated by ds_wrxchg_rtn_b32, see the match phase), the LINKS (u16 per position, a ring indexed by strip position mod
// kChainRing: the distance to the previous position with the same hash, 0: none), two exchange areas, then histogram,
// wave totals and the distance-symbol table as before.  The ring holds the window, the step whose chains are being
// walked and the step being inserted: a walk never reads a slot that the insertion beside it is writing.
constexpr uint32_t kChainRingSteps = kWindow / kStep + 2;
constexpr uint32_t kChainRing = kChainRingSteps * kStep;     // positions that have a link
constexpr uint32_t C_L_HEAD = L_TABLE;                       // u32[1<<kHashBits]
constexpr uint32_t C_L_PREV = C_L_HEAD + (4u << kHashBits);  // u16[kChainRing]
constexpr uint32_t C_L_POST = C_L_PREV + 2 * kChainRing;     // u16[2][kStep]: what the threads post for the serial pass, and what it answers
constexpr uint32_t C_L_HIST = C_L_POST + 4 * kStep;
constexpr uint32_t C_L_WTOT = C_L_HIST + 4 * kHistStride;
constexpr uint32_t C_L_DSYM = C_L_WTOT + 4 * K1_WAVES;
constexpr uint32_t C_K1_LDS = C_L_DSYM + 512;
static_assert(C_K1_LDS <= 160 * 1024 && C_L_PREV % 16 == 0 && C_L_HIST % 16 == 0 && C_L_WTOT % 16 == 0, "K1 (chains): one workgroup per CU");

// v_ffbl_b32 as the hardware defines it: the lowest set bit, 0xFFFFFFFF for 0 (spelled out, because the C
// builtins leave 0 undefined or make the compiler add a compare and a select to patch it)
__device__ __forceinline__ uint32_t ffbl(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// lowest set bit of x1:x0 (0..63), 0xFFFFFFFF when both are 0: the OR leaves the all-ones alone, the MIN skips it
__device__ __forceinline__ uint32_t first_bit64(uint32_t x0, uint32_t x1) { return min(ffbl(x0), ffbl(x1) | 32u); }

// first mismatching byte between the 8 bytes in a0,a1 and those at LDS byte address c + 4*WOFF (the word offset
// rides in the loads' immediate field); some huge value when all eight are equal
template <uint32_t WOFF>
__device__ __forceinline__ uint32_t cmp8(const uint32_t* d32, uint32_t a0, uint32_t a1, uint32_t c) {
  const uint32_t cw = c >> 2, csh = c & 3;
  const uint32_t c0 = d32[cw + WOFF], c1 = d32[cw + WOFF + 1], c2 = d32[cw + WOFF + 2];
  const uint32_t x0 = a0 ^ __builtin_amdgcn_alignbyte(c1, c0, csh);
  const uint32_t x1 = a1 ^ __builtin_amdgcn_alignbyte(c2, c1, csh);
  return first_bit64(x0, x1) >> 3;
}

// The same for ranking candidates: a candidate whose first four bytes differ can never become a match
// (kMinMatch = 4), so it ranks as 0 whatever its shorter common prefix is -- the result only has to be exact from 4 up.
// All dwords are loaded up front and pinned there (empty asm): left alone, the compiler sinks the third load
// of every candidate behind a branch on the first compare, which turns one LDS round trip per step into several
// dependent ones.
__device__ __forceinline__ uint32_t rank_of(uint32_t a0, uint32_t a1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t csh,
                                            uint32_t maxlen) {
  static_assert(kMinMatch == 4 && kRank == 8, "rank8: the first dword decides whether a candidate counts");
  const uint32_t x0 = a0 ^ __builtin_amdgcn_alignbyte(c1, c0, csh);
  const uint32_t x1 = a1 ^ __builtin_amdgcn_alignbyte(c2, c1, csh);
  // min(4 + equal bytes among 4..7, kRank, maxlen) as ONE three-way minimum (spelled out: the compiler rewrites
  // min(4 + x, 8) into 4 + min(x, 4) and then needs a second minimum); ffbl: 0xFFFFFFFF when bytes 4..7 are equal
  const uint32_t l = 4u + (ffbl(x1) >> 3);
  uint32_t r;
  asm("v_min3_u32 %0, %1, 8, %2" : "=v"(r) : "v"(l), "v"(maxlen));
  return x0 ? 0u : r;
}
// rank of the candidate at LDS byte address c, capped at maxlen
__device__ __forceinline__ uint32_t rank8(const uint32_t* d32, uint32_t a0, uint32_t a1, uint32_t c, uint32_t maxlen) {
  const uint32_t cw = (c & 0xFFFFu) >> 2;  // any c reads inside the LDS allocation (see the call site)
  uint32_t c0 = d32[cw], c1 = d32[cw + 1], c2 = d32[cw + 2];
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
  return rank_of(a0, a1, c0, c1, c2, c & 3, maxlen);
}
__device__ __forceinline__ void rank8x2(const uint32_t* d32, uint32_t a0, uint32_t a1, uint32_t ca, uint32_t cb,
                                        uint32_t maxlen, uint32_t& la, uint32_t& lb) {
  const uint32_t wa = (ca & 0xFFFFu) >> 2, wb = (cb & 0xFFFFu) >> 2;  // any address reads inside the LDS allocation
  uint32_t p0 = d32[wa], p1 = d32[wa + 1], p2 = d32[wa + 2], q0 = d32[wb], q1 = d32[wb + 1], q2 = d32[wb + 2];
  asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(q0), "+v"(q1), "+v"(q2));
  la = rank_of(a0, a1, p0, p1, p2, ca & 3, maxlen);
  lb = rank_of(a0, a1, q0, q1, q2, cb & 3, maxlen);
}

// SENSITIVITY PROBE (diagnostic builds only: -DSF_PROBE_STAGE / _MATCH / _PARSE / _EMIT=<n>, tools/exp/probe_phases.sh): n dead
// vector instructions (v_xor of a register with itself, the cheap issue class) in one phase of k_lz77 -- what a phase's
// time costs the kernel per instruction says whether taking instructions OUT of it can pay.  Compiled out otherwise.
template <int N>
__device__ __forceinline__ void probe_valu(uint32_t& r) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(r));
}
#ifndef SF_PROBE_STAGE
#define SF_PROBE_STAGE 0
#endif
#ifndef SF_PROBE_MATCH
#define SF_PROBE_MATCH 0
#endif
#ifndef SF_PROBE_PARSE
#define SF_PROBE_PARSE 0
#endif
#ifndef SF_PROBE_EMIT
#define SF_PROBE_EMIT 0
#endif

// Workgroup barrier that orders LDS traffic only: the match step keeps a global store (the position's distance, see
// k_lz77) in flight across its barriers, which __syncthreads() would wait for twice per step
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 16-bit step code -> position relative to the epoch's first step: (sc-1)*1024 + t with
// v = sc << 10 | (1023 - t); ordered like the specification's ((step+1) << 12) | (4095 - t), so MAX
// keeps the same winner (the first position of the latest step)
static_assert(kStep == 1024, "entry encoding");
__device__ __forceinline__ uint32_t entry_pos(uint32_t v) { return v - 1u - 2u * (v & 1023u); }
// entry_pos(v) + 1 in two instructions (the multiply-add is spelled out: left to itself the compiler expands the
// expression into twice as many shifts and masks); 0 for an empty half
template <uint32_t SH>
__device__ __forceinline__ uint32_t entry_rel(uint32_t v) {
  const uint32_t lo = v & ((1u << SH) - 1u);
  uint32_t c;
  asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(c) : "v"(lo), "v"(v));
  return c;
}

// STAMPS: diagnostic build only (SFH_K1_STAMPS=1), s_memtime at phase boundaries into `stamps`
// [strip][8] = cycles in {stage, match, take, walk, segpre, emit, flush}; never used for timing claims.
// DEPTH2: both history levels of a bucket are tried (effort 0); otherwise only the newer one (effort 1: a third
// fewer compares, about 3 % more output).  NEAR: the step-local candidate is tried too (efforts 0 and 1; effort 2
// leaves it out: one table read and one candidate fewer, another 2 % of output on text, more on repetitive data)
// STRIDE2: only the even positions are searched; an odd one takes over its successor's match when its own byte fits
// in front of it (efforts 0..2; SFH_EFFORT_THOROUGH searches every position).  The search is then pipelined over the
// two halves of the workgroup, see the match phase.
// LONG (SFH_EFFORT_MAX): the 32 KiB of table are TWO tables of 4096 buckets, one keyed by four bytes as ever, one by seven
// (kLongBytes): a position has four far candidates; of equally ranked ones the nearest wins.
// CHAIN (SFH_EFFORT_BEST / _ULTRA): exact hash chains instead of the step tables -- every position is inserted and
// searched, most recent candidate first, `chain_depth` of them at most (the specification's chain_depth; zlib's
// structure).  One workgroup per CU (the links take 80 KiB of LDS), 128 vector registers.
// RECENT (SFH_EFFORT_RECENT): the step tables with EXACT RECENCY -- see R_L_POST above and the match phase.
template <bool STAMPS, bool DEPTH2, bool NEAR, bool STRIDE2, bool LONG, bool CHAIN = false, bool RECENT = false>
__global__ __launch_bounds__(K1_THREADS, CHAIN ? K1_THREADS / 256 : 2 * K1_THREADS / 256) void k_lz77(
    const uint8_t* __restrict__ src, uint64_t n_total, uint32_t strip_bytes, uint16_t* __restrict__ items,
    uint32_t* __restrict__ nitems_out, uint32_t* __restrict__ ntok_out, uint32_t* __restrict__ hist_out,
    uint32_t* __restrict__ rtok_out, uint32_t lazy, uint32_t fast_skip, uint64_t* __restrict__ stamps,
    uint32_t chain_depth) {
  static_assert(!CHAIN || (!STRIDE2 && !LONG && DEPTH2 && NEAR), "chains: every position searched, no step tables");
  static_assert(!RECENT || (!LONG && !CHAIN && DEPTH2 && NEAR), "recent: the step tables' search patterns (stride 2 or every position)");
  constexpr uint32_t LH = CHAIN ? C_L_HIST : L_HIST, LW = CHAIN ? C_L_WTOT : L_WTOT, LD = CHAIN ? C_L_DSYM : L_DSYM;
  uint64_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t st_t = 0;
  auto stamp = [&](int slot) {
    if constexpr (STAMPS) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      st_acc[slot] += now - st_t;
      st_t = now;
    }
  };
  if constexpr (STAMPS) st_t = __builtin_amdgcn_s_memtime();
  // positions per insertion step: 1024 of which the even ones are searched, or 512 all of which are (thorough)
  constexpr uint32_t SH = (STRIDE2 || CHAIN) ? 10u : 9u, STEP = 1u << SH;
  constexpr uint32_t HB = LONG ? kHashBits - 1 : kHashBits;  // bucket index bits of a table
  static_assert(!LONG || (DEPTH2 && !STRIDE2), "the second table comes with the highest effort");
  // step codes in a 16-bit table half: ((step - epoch) + 1) << SH | (STEP - 1 - index in the step); the epoch advances by
  // kEpS steps whenever a round would reach kEpMax steps past it, entries older than that vanish
  constexpr uint32_t kEpS = (kEpochSteps << 10) >> SH, kEpMax = (kEpochMax << 10) >> SH;
  static_assert(((kEpMax + 1) << SH) <= 65536 && kRound % STEP == 0, "step codes");

  // static LDS (its address is a compile-time constant: no base register, no add per access)
  __shared__ __attribute__((aligned(16))) uint8_t smem[CHAIN ? C_K1_LDS : RECENT ? R_K1_LDS : K1_LDS];
  uint32_t* s_data = reinterpret_cast<uint32_t*>(smem + L_DATA);
  uint8_t* s_bytes = smem + L_DATA;
  uint32_t* s_len4 = reinterpret_cast<uint32_t*>(smem + L_LEN4);
  uint32_t* s_table = reinterpret_cast<uint32_t*>(smem + L_TABLE);
  [[maybe_unused]] uint32_t* s_table2 = s_table + (1u << HB);  // LONG: the seven-byte table behind the four-byte one
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + LH);
  uint32_t* s_wtot = reinterpret_cast<uint32_t*>(smem + LW);
  [[maybe_unused]] uint32_t* s_head = reinterpret_cast<uint32_t*>(smem + C_L_HEAD);
  [[maybe_unused]] uint16_t* s_prev = reinterpret_cast<uint16_t*>(smem + C_L_PREV);

  const uint32_t t = threadIdx.x;
  // t >> 6 is wave-uniform, but only readfirstlane tells the compiler so
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6)), lane = t & 63;
  const uint32_t strip = blockIdx.x;
  const uint64_t sbase = (uint64_t)strip * strip_bytes;
  const uint32_t n = (uint32_t)((n_total - sbase) < (uint64_t)strip_bytes ? (n_total - sbase) : strip_bytes);
  const uint32_t chunk0 = strip * (strip_bytes / kChunk);
  const uint8_t* const sp = src + sbase;
  // @phase inherit
  // four bytes of the strip at `pos` (multiple of 4), zero beyond n
  auto load4 = [&](uint32_t pos) -> uint32_t {
    if (pos + 4 <= n) return *reinterpret_cast<const uint32_t*>(sp + pos);
    // @phase +tail trips=0 note=the strip's last bytes, byte by byte
    uint32_t w = 0;
    for (uint32_t b = 0; pos + b < n; ++b) w |= (uint32_t)sp[pos + b] << (8 * b);
    return w;
    // @phase inherit
  };

  // ---- prologue: empty table and histogram; the strip's first kLook bytes where the first shift finds them ----
  {
    uint4* t4 = reinterpret_cast<uint4*>(smem + L_TABLE);
    for (uint32_t idx = t; idx < (1u << kHashBits) / 4; idx += K1_THREADS) t4[idx] = make_uint4(0, 0, 0, 0);
    for (uint32_t idx = t; idx < kHistStride; idx += K1_THREADS) s_hist[idx] = (idx == 256) ? 1u : 0u;
    if (t < kLook / 4) s_data[(kWindow + kRound) / 4 + t] = load4(4 * t);
    if (t < 4) s_len4[kRound / 8 + t] = 0;  // pad read by the take pass
    if (t < 512) smem[LD + t] = (uint8_t)dist_symbol_of(t < 256 ? t : (t - 256) << 7);
  }
  // this thread's eight bytes of round 0 (positions kLook + 8t ..)
  uint32_t pre_lo = load4(kLook + 8 * t), pre_hi = load4(kLook + 8 * t + 4);
  __syncthreads();
  stamp(0);

  const uint32_t nrounds = (n + kRound - 1) / kRound;
  // First step of the table's epoch (uniform, mod 2^32).  It starts a window's worth of steps BEFORE the strip and
  // stays at least that far behind the step in hand (ageing moves it by kEpS when it is kEpMax behind): a position is
  // never less than kWindow past the epoch's start, so "within the window" is one compare of the coded
  // position with a threshold, and an empty half (code 0 = the last position of the step before the epoch) fails it
  // like any entry that is too old
  static_assert(kEpMax - kEpS >= kWindow / STEP, "the epoch stays a window behind");
  uint32_t ebase = 0u - kWindow / STEP;
  uint32_t tot_tok = 0, tot_items = 0;   // of the chunk in flight (uniform)
  bool skip = false;                     // stored fast path (uniform): the chunk in hand took it (decided behind its first round's parse)
  bool short_probe = false;              // (uniform) the strip's previous chunk took it: only kSkipProbe positions of this one's span are searched
  bool lds_stale = false;                // (uniform) the rounds behind a chunk's probe round were taken in one go: the window in LDS is not the strip's
  uint32_t stale_span = 0;               // ... and the positions of that chunk that were searched and inserted (its probe span)

  // @phase round.head trips=1 note=round bookkeeping
  for (uint32_t r = 0; r < nrounds; ++r) {
    const uint32_t rb = r * kRound;                         // strip position of the round's first byte
    const uint32_t rc = r % kRoundsPerChunk;                // round within its chunk
    const uint32_t chunk = chunk0 + r / kRoundsPerChunk;
    const uint32_t cstart = rb - rc * kRound;               // strip position of the chunk's first byte
    const uint32_t qn = (n - rb) < kRound ? (n - rb) : kRound;  // valid positions in this round
    uint16_t* const gi = items + (uint64_t)chunk * kChunk;
    if (rc == 0) {
      short_probe = skip && fast_skip && (n - cstart) > kSkipSpan;  // (skip: still the previous chunk's)
      tot_tok = 0; tot_items = 0; skip = false;
    }
    // (the stored fast path is decided at the end of a chunk's first round, behind its parse: skip_now below)

    bool chunk_done = false;  // (uniform) stored fast path: the chunk's later rounds were taken together with this one (skip_now)
    // RECENT: the histogram's place holds the match phase's posts; its counts wait here meanwhile
    [[maybe_unused]] uint32_t hsave = 0;
    // @phase stage trips=1
    // ---- stage: shift the window down by one round, append the prefetched 4 KiB ----
    {
      uint4* s4 = reinterpret_cast<uint4*>(smem + L_DATA);
      constexpr uint32_t kUnits = (kWindow + kLook) / 16, kOff = kRound / 16;  // 2050 units move down by 512
      static_assert(kUnits > 2 * K1_THREADS && kUnits <= 3 * K1_THREADS, "shift: three units per thread");
      const bool reload = lds_stale;  // (uniform)
      if (reload) {
        // behind a chunk that took the stored fast path (its later rounds never came through here): what the window must hold
        // and this round come from the input again instead of from three shifts per skipped chunk.  Of the window only the
        // skipped chunk's PROBE SPAN was ever inserted into the table (stale_span positions at the window's start, and
        // anything older is farther back than 32 KiB: rejected by distance, its bytes read and dropped like any outdated
        // entry's), so only that span and what a match starting in it can reach (258 bytes + the compares' slack) are
        // loaded; the rest of the window stays whatever it is.  (rb >= kChunk: the skipped chunk lies in this strip.  Every
        // reader of the old LDS bytes is behind the barrier at the skipped chunk's end.)
        constexpr uint32_t kReach = 320, kRoundUnits = (kRound + kLook) / 16;
        static_assert(kReach >= 258 + kCap + 32 && (kSkipSpan + kReach) / 16 < K1_THREADS - 1 && kRoundUnits < K1_THREADS - 1, "reload: two units per thread, the last thread a third");
        const uint32_t wunits = (stale_span + kReach) / 16;
        const uint8_t* const wp = sp + (rb - kWindow);
        // (... and the window's LAST bytes: the byte in front of the round's first position is looked at whenever a match
        // starts there -- the odd neighbour's byte check of a near candidate, the run test of the wave-wide extension)
        if (rb + kRound + kLook <= n) {
          if (t < wunits) s4[t] = reinterpret_cast<const uint4*>(wp)[t];
          if (t < kRoundUnits) s4[kWindow / 16 + t] = reinterpret_cast<const uint4*>(wp)[kWindow / 16 + t];
          if (t == K1_THREADS - 1) s4[kWindow / 16 - 1] = reinterpret_cast<const uint4*>(wp)[kWindow / 16 - 1];
        } else {  // the strip ends inside: word by word, zeros beyond the end
          for (uint32_t i = t; i < 4 * wunits; i += K1_THREADS) s_data[i] = load4(rb - kWindow + 4 * i);
          for (uint32_t i = t; i < 4 * kRoundUnits; i += K1_THREADS) s_data[kWindow / 4 + i] = load4(rb + 4 * i);
          if (t == K1_THREADS - 1) s_data[kWindow / 4 - 1] = load4(rb - 4);
        }
        lds_stale = false;
      } else {
        const uint4 c0 = s4[kOff + t], c1 = s4[kOff + K1_THREADS + t];
        uint4 c2 = make_uint4(0, 0, 0, 0);
        if (t < kUnits - 2 * K1_THREADS) c2 = s4[kOff + 2 * K1_THREADS + t];
        __syncthreads();  // every read of the old window (this shift, the previous round's emit) precedes the writes
        s4[t] = c0;
        s4[K1_THREADS + t] = c1;
        if (t < kUnits - 2 * K1_THREADS) s4[2 * K1_THREADS + t] = c2;
      }
      if (!reload) *reinterpret_cast<uint2*>(&s_data[(kWindow + kLook) / 4 + 2 * t]) = make_uint2(pre_lo, pre_hi);
      if constexpr (SF_PROBE_STAGE > 0) { uint32_t pr = t; probe_valu<SF_PROBE_STAGE>(pr); }
      // (RECENT asks for them behind the match phase, whose serial pass wants the registers: parse and emit hide the latency)
      if (!RECENT && r + 1 < nrounds) {
        pre_lo = load4(rb + kRound + kLook + 8 * t);
        pre_hi = load4(rb + kRound + kLook + 8 * t + 4);
      }
      // age the step codes: all move down by kEpochSteps steps, saturating at 0 -- ONE packed 16-bit instruction per
      // bucket (v_pk_sub_u16 clamp).  What falls below 1 << 10 (step field 0) is older than the window: empty
      // @phase stage.ageing trips=1 depth=2 note=two iterations every second round
      if (rb / STEP - ebase >= kEpMax) {
        // (behind a chunk whose later rounds were taken in one go the epoch is further back than one move mends: the moves
        // that were due meanwhile are made up in this one -- at most three, 3 * kEpS << SH < 2^16; what ageing removes is
        // older than the window either way, so when it happens shows nowhere)
        const uint32_t moves = (rb / STEP - ebase - kEpMax) / kEpS + 1u;
        static_assert(3 * (kEpS << SH) < 65536 && (kRoundsPerChunk * (kRound / STEP) - 1) / kEpS + 1 <= 3, "ageing: one packed subtract makes up for a skipped chunk");
        const uint32_t kSub2 = moves * ((kEpS << SH) * 0x00010001u);
        uint4* t4 = reinterpret_cast<uint4*>(smem + L_TABLE);
        static_assert((1u << kHashBits) % (4 * K1_THREADS) == 0, "ageing: whole 16-byte units per thread");
        for (uint32_t idx = t; idx < (1u << kHashBits) / 4; idx += K1_THREADS) {
          uint4 e = t4[idx];
          asm("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(e.x) : "s"(kSub2));
          asm("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(e.y) : "s"(kSub2));
          asm("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(e.z) : "s"(kSub2));
          asm("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(e.w) : "s"(kSub2));
          t4[idx] = e;
        }
        ebase += moves * kEpS;
      }
      // @phase stage.end trips=1 note=closing barrier
      // (the barrier at the top of the stage follows the previous round's last histogram update)
      if constexpr (RECENT) hsave = t < kHistStride ? s_hist[t] : 0u;
      __syncthreads();
    }
    stamp(0);

    {  // (a round: match, parse, emit)
      // @phase match.setup trips=1
      // ---- match finding over this round ----
      uint32_t nsteps = (qn + STEP - 1) / STEP;
      if (short_probe && rc == 0) {
        // behind a chunk that took the stored fast path: kSkipProbe positions are searched, the rest of the span has no match
        static_assert(kSkipProbe % STEP == 0 && kSkipProbe < kSkipSpan && kSkipSpan == kRound, "short probe: whole steps of the chunk's first round");
        nsteps = nsteps < kSkipProbe / STEP ? nsteps : kSkipProbe / STEP;
        for (uint32_t idx = kSkipProbe / 8 + t; idx < kRound / 8; idx += K1_THREADS) s_len4[idx] = 0;  // (the steps that run write below this)
      }
      const uint32_t K = kWindow + ebase * STEP - rb;  // LDS byte address of a coded position = entry_pos(code) + K (mod 2^32)
      // The CU's two workgroups are in different phases most of the time.  The match phase is the long one and the one
      // that keeps the LDS and the vector units busy, so its waves go first when both workgroups have instructions ready
      // (measured: match 2 > walk 1 > stage = emit 0 takes 4 % off the kernel; walk at or above match gives it all back)
      // The round's distances (16 bits per position) do not live in LDS: they are STAGED in the chunk's own item array,
      // in the slots of the round's positions -- free at this point, because a chunk never has more items than positions
      // (a match of kMinMatch bytes or more takes two) -- and every lane reads its own back at the start of the parse,
      // before any item of the round is written.  The 16 KiB this frees in LDS hold the larger hash table.  (With the even
      // positions searched only, one slot per even position: an odd position's distance is its successor's.)
      // WHERE in the item array: right behind the items the chunk has so far, rounded up to a 128-byte line (round 5; before:
      // at the round's own positions, 2 * rc * kRound).  The round's items then land on the very lines the staged distances
      // dirtied, a few thousand cycles later, and the L2 writes such a line back once -- at the positions' own place the
      // staging of a chunk's later rounds lay beyond anything its items ever reach and went out to HBM as it was: WRITE_SIZE
      // of the kernel 1.81 -> 1.10 GB per GiB of text.  Whole LINES: the same offset rounded to 16 bytes only made the kernel
      // FETCH 0.64 GB more (1.88 against 1.24 GB: lines shared by the previous round's items and this round's slots).
      // Never beyond the old offset + what rounding adds inside the array (a chunk has no more items than positions, and
      // 2 * 3 * kRound is a multiple of 128)
      const uint32_t stage_off = (2u * tot_items + 127u) & ~127u;  // byte offset of the round's slots in the chunk's item array
      __builtin_amdgcn_s_setprio(2);
      // @phase match.chain trips=0
      if constexpr (CHAIN) {
        // ---- exact hash chains (the specification's chain_depth > 0) ----
        // A step is 1024 positions, one per thread.  INSERTION is serial in the position -- a position's link is the latest
        // EARLIER position with its hash -- and one LDS instruction does it for 64 positions at once: ds_wrxchg_rtn_b32
        // executes the lanes of a wave-instruction in ASCENDING LANE ORDER where they meet at one address
        // (tools/micro/lds_xchg_order.hip: 472 M lane observations on every CU, no exception; tests/test_gpu_parity.py runs
        // it as a guard), so `old = exchange(head[hash], own code)` hands every lane the nearest lower lane with its hash,
        // or what the head held before the slice, and leaves the slice's last position of every hash in the head.  Every
        // thread posts its hash; ONE wave (they take turns) then issues the sixteen slices' exchanges in order, back to
        // back -- the LDS executes a wave's operations in the order they were issued, so no barrier is needed between
        // slices -- while the others WALK the chains of the step before, whose links are complete (the links a walk follows
        // are older than anything being inserted).  Two barriers per step.
        static_assert(STEP == K1_THREADS && K1_WAVES == 16, "one position per thread and step, sixteen slices");
        uint16_t* const s_post = reinterpret_cast<uint16_t*>(smem + C_L_POST);  // [STEP] hash | inserted << 15
        uint16_t* const s_hv = s_post + STEP;                                    // [STEP] the head's code before the position went in
        static_assert(kHashBits <= 15, "posted entry layout");
        const uint32_t ring0 = (r * (kRound / STEP)) % kChainRingSteps;  // (uniform) ring step of the round's first step
        uint8_t* const gst = reinterpret_cast<uint8_t*>(gi) + stage_off;
        const uint32_t sh0 = t & 3u;
        const uint32_t to_rend = kRegion - (t & (kRegion - 1));
        const uint32_t mlen0 = to_rend < kCap ? to_rend : kCap;
        // the walk of the position this thread owns in the step before
        uint32_t w_a0 = 0, w_a1 = 0, w_a2 = 0, w_a3 = 0, w_maxlen = 0, w_tot = 0, w_best = 0, w_bdist = 0, w_q = 0, w_idx = 0;
        bool w_act = false;
        // One candidate of a walk: the 16 bytes at distance `tot` behind the position (five dwords, whatever its alignment)
        // and its own link.  (Measured and not kept: the same bytes as three aligned ds_read_b64 + selects, 21.8 against
        // 19.9 ms per GiB at depth 8 -- random 8-byte gathers cost the LDS more than their bank model says, as in the
        // table path; and the second eight bytes fetched only where the first eight are equal: no faster.)  A lane that has nothing to look at asks for distance 0 -- its own position -- and ignores the
        // answer, so the loads are straight-line code and the next candidate's can be in flight while this one is compared:
        // a walk is a chain of dependent LDS round trips (link -> address -> link), the comparing is not on it.
        struct Cand { uint32_t p0, p1, p2, p3, p4, dn, csh; };
        auto fetch = [&](uint32_t tot) -> Cand {
          const uint32_t c = (kWindow + w_q) - tot;            // the candidate's LDS byte address (inside the window)
          int32_t ci = (int32_t)w_idx - (int32_t)tot;
          ci += (ci >> 31) & (int32_t)kChainRing;              // ... and its ring index
          const uint32_t cw = c >> 2;
          Cand k;
          k.dn = s_prev[ci];                                   // (asked for first: the next address only waits for this one)
          k.p0 = s_data[cw]; k.p1 = s_data[cw + 1]; k.p2 = s_data[cw + 2]; k.p3 = s_data[cw + 3]; k.p4 = s_data[cw + 4];
          k.csh = c & 3u;
          return k;
        };
        auto walk_chain = [&]() {
          uint32_t tot = w_act ? w_tot : 0u;
          bool act = w_act;
          Cand cur = fetch(tot);
          for (uint32_t k = 0; k < chain_depth; ++k) {
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            const uint32_t nt = tot + cur.dn;
            const bool actn = act && cur.dn != 0 && nt <= kWindow;  // the chain goes on, inside the window
            const Cand nxt = fetch(actn ? nt : 0u);
            const uint32_t x0 = w_a0 ^ __builtin_amdgcn_alignbyte(cur.p1, cur.p0, cur.csh), x1 = w_a1 ^ __builtin_amdgcn_alignbyte(cur.p2, cur.p1, cur.csh);
            const uint32_t x2 = w_a2 ^ __builtin_amdgcn_alignbyte(cur.p3, cur.p2, cur.csh), x3 = w_a3 ^ __builtin_amdgcn_alignbyte(cur.p4, cur.p3, cur.csh);
            // first differing bit of x3:x2:x1:x0 (all ones when there is none: ffbl(0) = 0xFFFFFFFF survives the ORs)
            const uint32_t fb = min(min(ffbl(x0), ffbl(x1) | 32u), min(ffbl(x2) | 64u, ffbl(x3) | 96u));
            const uint32_t l = min(fb >> 3, w_maxlen);         // equal bytes, kCap and the region's end at most
            if (act && l > w_best) { w_best = l; w_bdist = tot; }  // the longest; the first found (the nearest) on ties
            act = actn && w_best < kCap;                       // (nothing later can beat a capped match: it would only tie)
            tot = actn ? nt : 0u;
            cur = nxt;
          }
        };
        // [A] of a step: the position's bytes and hash, posted for the serial pass.  It runs one stage AHEAD (for step 0 before
        // the loop, for step it + 1 at the end of iteration it), so a step costs two barriers, not three: the barrier that
        // makes a step's links visible is also the one that makes the next step's posts visible.
        uint32_t n_a0 = 0, n_a1 = 0, n_a2 = 0, n_a3 = 0;
        bool n_ins = false;
        auto stage_a = [&](uint32_t st) {                      // st: (uniform) step index in the round, < nsteps
          const uint32_t q = st * STEP + t;
          const uint32_t wb = kWindow + (q & ~3u);
          const uint32_t d0 = *reinterpret_cast<const uint32_t*>(smem + wb), d1 = *reinterpret_cast<const uint32_t*>(smem + wb + 4),
                         d2 = *reinterpret_cast<const uint32_t*>(smem + wb + 8), d3 = *reinterpret_cast<const uint32_t*>(smem + wb + 12),
                         d4 = *reinterpret_cast<const uint32_t*>(smem + wb + 16);
          n_a0 = __builtin_amdgcn_alignbyte(d1, d0, sh0); n_a1 = __builtin_amdgcn_alignbyte(d2, d1, sh0);
          n_a2 = __builtin_amdgcn_alignbyte(d3, d2, sh0); n_a3 = __builtin_amdgcn_alignbyte(d4, d3, sh0);
          const uint32_t h = (n_a0 * 2654435761u) >> (32 - kHashBits);
          n_ins = rb + q + kMinMatch <= n;                     // the specification inserts and searches what has four bytes left
          s_post[t] = (uint16_t)(h | (n_ins ? 1u << 15 : 0u));
        };
        if (nsteps) stage_a(0);
        lds_barrier();
        for (uint32_t it = 0; it <= nsteps; ++it) {
          const uint32_t sb = it * STEP;                       // (uniform) the step's first position, round-relative
          const uint32_t q = sb + t;
          const bool ins = it < nsteps && n_ins;
          const uint32_t a0 = n_a0, a1 = n_a1, a2 = n_a2, a3 = n_a3;
          const uint32_t code = ((rb / STEP + it - ebase + 1) << SH) | t;  // step code | index in the step: ascending in the position
          if (it < nsteps && wave == (it & (K1_WAVES - 1))) {  // (uniform) this wave's turn: the step goes into the heads, slice by slice
            // Nothing here waits for an answer before the next request goes out: the sixteen posted entries are fetched
            // together, the sixteen exchanges follow one another, the answers are stored at the end
            const uint32_t code0 = ((rb / STEP + it - ebase + 1) << SH) | lane;
            uint32_t e[K1_WAVES], old[K1_WAVES];
#pragma unroll
            for (uint32_t sl = 0; sl < K1_WAVES; ++sl) e[sl] = s_post[sl * 64 + lane];
#pragma unroll
            for (uint32_t sl = 0; sl < K1_WAVES; ++sl) {
              old[sl] = 0;
              if (e[sl] & (1u << 15)) old[sl] = atomicExch(&s_head[e[sl] & ((1u << kHashBits) - 1u)], code0 + sl * 64);
            }
#pragma unroll
            for (uint32_t sl = 0; sl < K1_WAVES; ++sl) s_hv[sl * 64 + lane] = (uint16_t)old[sl];
          }
          if (it >= 1) walk_chain();
          lds_barrier();
          const uint32_t hv = ins ? s_hv[t] : 0u;
          if (it >= 1) {
            // ---- the position of the step before is settled: its length and distance go where the parse finds them ----
            const uint32_t bd1 = w_bdist - 1u;                 // distance - 1 (what is staged; only read where there is a match)
            const bool ok = w_best >= (bd1 >= kFar4 ? kMinMatch + 1 : kMinMatch);
            uint32_t v = ok ? w_best - 3u : 0u;
            *reinterpret_cast<uint16_t*>(gst + 2u * w_q) = (uint16_t)bd1;
            // two lanes' 4-bit lengths -> one byte, stored by the even lanes (see the thorough path below)
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true) << 4;
            asm volatile("s_mov_b64 exec, %2\n\tds_write_b8 %0, %1 offset:%3\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                         :: "v"(w_q >> 1), "v"(v), "s"(0x5555555555555555ull), "n"(L_LEN4) : "memory");
          }
          if (it < nsteps) {
            // the position's link: the distance to what the head named before it went in -- an earlier position of this
            // step or any before; an empty head, or one aged out (older than the window anyway): none
            const uint32_t d = (ins && hv >= STEP) ? code - hv : 0u;
            uint32_t rs = ring0 + it;                          // (uniform) the step's place in the ring
            rs -= rs >= kChainRingSteps ? kChainRingSteps : 0u;
            w_idx = rs * STEP + t;
            s_prev[w_idx] = (uint16_t)d;
            w_a0 = a0; w_a1 = a1; w_a2 = a2; w_a3 = a3; w_q = q;
            w_maxlen = (uint32_t)max(min((int)(qn - sb) - (int)t, (int)mlen0), 0);
            w_tot = d; w_best = 0; w_bdist = 0;
            w_act = d != 0 && d <= kWindow && w_maxlen >= kMinMatch;
            if (it + 1 < nsteps) stage_a(it + 1);              // (the posts of this step were read before the barrier above)
          }
          lds_barrier();  // the step's links and the next step's posts are in place
        }
      } else if constexpr (RECENT) {
        // ---- exact recency (the specification's `recent`, near_depth 1, link_steps 1) ----
        // The step tables' search pattern -- the even positions of a step of 1024 searched and every position inserted
        // (STRIDE2), or every one of a step's 512 positions searched; a search split over two intervals and the two halves
        // of the workgroup -- with other candidates.  A bucket is {hi, lo}: lo the LATEST position with the hash, hi what
        // lo held before the most recent step that inserted the hash.  A position's candidates are lo and hi as they
        // stand BEFORE its step and its exact PREDECESSOR, the nearest earlier position with its hash, in the step or
        // before it.  Insertion is `old = {hi, lo} <- {hi, own code}` in ascending position order: ds_mskor_rtn_b32 with
        // mask 0xFFFF replaces the low half alone and returns the bucket as it was, and the LDS executes the lanes of
        // such a wave-instruction in ascending order where they meet at one bucket (like the chains' exchange;
        // sf_guard.hip checks it on the device at hand), so old.lo IS the predecessor.  ONE wave issues the step's slices
        // in order, back to back.  What it needs, the hashes, the searching half POSTS before barrier 1; what comes
        // back is the ANSWER a searched position reads an interval later.  Between the barriers the posting half stores
        // the buckets' new hi (the lo it read before the step: every position with the hash stores the same) and ranks
        // its far candidates, and one wave of the other half does the pass -- the pass is not alone on the critical path.
        // Step codes ASCEND with the position here: ((step - epoch) + 1) << SH | index in the step.
        constexpr uint32_t NSL = STEP / 64;               // slices of a step
        static_assert((NSL == 16 || NSL == 8) && kHashBits <= 16 && 2 * STEP <= 4 * kHistStride && 2 * 512 <= kStep,
                      "one posted u16 per position, one answer per search");
        uint16_t* const s_post = reinterpret_cast<uint16_t*>(smem + R_L_POST);  // [STEP] hash of every position of the step
        uint16_t* const s_ans = reinterpret_cast<uint16_t*>(smem + R_L_ANS);    // [512] the bucket's lo before a searched position went in
        const uint32_t grp = wave >> 3;                   // (uniform) this wave's half
        const uint32_t tp = t & 511u;                     // index in the half
        const uint32_t ps = STRIDE2 ? 2 * tp : tp;        // the thread's searched position within a step
        const uint32_t to_rend = kRegion - (ps & (kRegion - 1));
        [[maybe_unused]] const bool inh_here = tp != 0 && (tp & (kRegion / 2 - 1)) != 0;
        [[maybe_unused]] const uint32_t first_ad = kWindow - rb;
        const uint32_t sh0 = ps & 3u;
        const uint32_t pswK = kWindow + (ps & ~3u);
        const uint32_t psK1 = kWindow - 1u + ps;
        const uint32_t mlen0 = to_rend < kCap ? to_rend : kCap;
        auto lds32 = [&](uint32_t at, uint32_t off) { return *reinterpret_cast<const uint32_t*>(smem + at + off); };
        uint8_t* const gst = reinterpret_cast<uint8_t*>(gi) + stage_off;
        const uint32_t KK = K - STEP;                     // LDS byte address of a coded position = code + KK (mod 2^32)
        uint32_t f_a0 = 0, f_a1 = 0, f_m0 = 0, f_m1 = 0, f_q0 = 0, f_q1 = 0, f_maxlen = 0;
        for (uint32_t it = 0; it <= nsteps; ++it) {
          const uint32_t code_it = (rb / STEP + it - ebase + 1) << SH;  // (uniform) step code of step `it`
          const bool first_half = (it & 1) == grp;        // (uniform) this half searches step `it`; the other finishes step it - 1
          uint32_t farv = 0, f_h = 0;
          [[maybe_unused]] uint32_t farv2 = 0;
          if (first_half) {
            if (it < nsteps) {
              // ---- first part (a): bytes, the hashes posted, the buckets as they stand before the step ----
              const uint32_t wb = it * STEP + pswK;
              const uint32_t d0 = lds32(wb, 0), d1 = lds32(wb, 4), d2 = lds32(wb, 8);
              f_a0 = __builtin_amdgcn_alignbyte(d1, d0, sh0);
              f_a1 = __builtin_amdgcn_alignbyte(d2, d1, sh0);
              const uint32_t h = (f_a0 * 2654435761u) >> (32 - kHashBits);
              farv = s_table[h];
              if constexpr (STRIDE2) {
                // the odd position behind this one is only inserted: its four bytes are in the registers already
                const uint32_t h2 = (__builtin_amdgcn_alignbyte(f_a1, f_a0, 1) * 2654435761u) >> (32 - kHashBits);
                farv2 = s_table[h2];
                f_h = h | (h2 << 16);
                *reinterpret_cast<uint32_t*>(smem + R_L_POST + 4u * tp) = f_h;  // positions ps and ps + 1
              } else {
                f_h = h;
                s_post[tp] = (uint16_t)h;
              }
            }
          } else if (it >= 1) {
            // ---- second part of the search at position ps of step it - 1 ----
            const uint32_t sb = (it - 1) * STEP;
            const uint32_t ad1 = sb + psK1;
            const uint32_t wb = sb + pswK;
            const uint32_t a0 = f_a0, a1 = f_a1, maxlen = f_maxlen;
            uint32_t neare = s_ans[tp];
            uint32_t pbyte = STRIDE2 ? smem[ad1] : 0u;
            uint32_t d2 = lds32(wb, 8), d3 = lds32(wb, 12), d4 = lds32(wb, 16);
            asm volatile("" : "+v"(neare), "+v"(pbyte), "+v"(d2), "+v"(d3), "+v"(d4));  // one round trip for all
            // the exact predecessor: an earlier position by construction; inside the window?
            const int32_t thr = (int32_t)((sb + ps) - KK);
            const uint32_t nc = neare + KK;
            const bool okn = (int32_t)neare >= thr;
            const uint32_t ln = rank8(s_data, a0, a1, nc, maxlen);
            // longest wins; ties go to the smaller distance: the predecessor, then lo, then hi
            uint32_t best = okn ? ln : 0u, bq = nc;
            if (f_m0 > best) { best = f_m0; bq = f_q0; }
            if (f_m1 > best) { best = f_m1; bq = f_q1; }
            const uint32_t bd1 = ad1 - bq;
            const uint32_t cbyte = STRIDE2 ? s_bytes[(bq - 1) & 0xFFFFu] : 0u;
            if (best == kRank) {
              const uint32_t a2 = __builtin_amdgcn_alignbyte(d3, d2, sh0), a3 = __builtin_amdgcn_alignbyte(d4, d3, sh0);
              const uint32_t lx = kRank + cmp8<kRank / 4>(s_data, a2, a3, bq);
              best = lx < maxlen ? lx : maxlen;
            }
            const bool ok = best >= (bd1 >= kFar4 ? kMinMatch + 1 : kMinMatch);
            const uint32_t len4 = ok ? best - 3 : 0u;
            if constexpr (STRIDE2) {
              const bool inh = ok && inh_here && pbyte == cbyte && bq != first_ad;
              const uint32_t len4o = inh ? (best < kCap ? best - 2 : kCap - 3) : 0u;
              *reinterpret_cast<uint16_t*>(gst + (ad1 - (kWindow - 1u))) = (uint16_t)bd1;
              smem[L_LEN4 + ((sb >> 1) + tp)] = (uint8_t)(len4o | (len4 << 4));
            } else {
              *reinterpret_cast<uint16_t*>(gst + 2u * (ad1 - (kWindow - 1u))) = (uint16_t)bd1;
              uint32_t v = len4;
              v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true) << 4;
              asm volatile("s_mov_b64 exec, %2\n\tds_write_b8 %0, %1 offset:%3\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                           :: "v"((sb + tp) >> 1), "v"(v), "s"(0x5555555555555555ull), "n"(L_LEN4) : "memory");
            }
          }
          if (it == nsteps) break;
          lds_barrier();  // the posts are in place, every read of the table as it stands before step `it` is done
          if (first_half) {
            // ---- first part (b): the buckets' new hi; the far candidates lo and hi, ranked ----
            *reinterpret_cast<uint16_t*>(smem + L_TABLE + 2u + 4u * (f_h & 0xFFFFu)) = (uint16_t)farv;
            if constexpr (STRIDE2) *reinterpret_cast<uint16_t*>(smem + L_TABLE + 2u + 4u * (f_h >> 16)) = (uint16_t)farv2;
            const uint32_t sb = it * STEP;
            const uint32_t m0 = farv & 0xFFFFu, m1 = farv >> 16;
            const uint32_t c0 = m0 + KK, c1 = m1 + KK;
            // ad - c <= kWindow  <=>  code >= thr, as signed numbers (thr > STEP: an empty half, or what ageing left of
            // an entry, lies below it)
            const int32_t thr = (int32_t)((sb + ps) - KK);
            const bool ok0 = (int32_t)m0 >= thr, ok1 = (int32_t)m1 >= thr;
            const uint32_t maxlen = (uint32_t)max(min((int)((qn - sb) - ps), (int)mlen0), 0);
            uint32_t l0, l1;
            rank8x2(s_data, f_a0, f_a1, c0, c1, maxlen, l0, l1);
            f_maxlen = maxlen;
            f_q0 = c0; f_q1 = c1;
            f_m0 = ok0 ? l0 : 0u;
            f_m1 = ok1 ? l1 : 0u;
          } else if ((wave & 7u) == 0) {
            // ---- the step goes into the buckets, slice by slice, in position order (one wave of the idle half) ----
            // (Positions past the input's end go in like the rest: they follow every position that is searched.)
            __builtin_amdgcn_s_setprio(3);
            uint32_t o[NSL];  // the bucket's byte offset in the table, then what the bucket held
#pragma unroll
            for (uint32_t sl = 0; sl < NSL; ++sl) o[sl] = (uint32_t)s_post[sl * 64 + lane] << 2;
            uint32_t dat = code_it | lane;
            const uint32_t msk = 0xFFFFu;
            // ONE asm statement from the first atomic to the wait: between two statements the compiler could move or spill a
            // register whose value the LDS has not delivered yet
#define SF_MSKOR(O) "ds_mskor_rtn_b32 " O ", " O ", %[m], %[d] offset:%[tab]\n\tv_add_u32 %[d], 64, %[d]\n\t"
#define SF_MSKOR8 SF_MSKOR("%0") SF_MSKOR("%1") SF_MSKOR("%2") SF_MSKOR("%3") SF_MSKOR("%4") SF_MSKOR("%5") SF_MSKOR("%6") SF_MSKOR("%7")
#define SF_MSKOR8B SF_MSKOR("%8") SF_MSKOR("%9") SF_MSKOR("%10") SF_MSKOR("%11") SF_MSKOR("%12") SF_MSKOR("%13") SF_MSKOR("%14") SF_MSKOR("%15")
            if constexpr (NSL == 16) {
              asm volatile(SF_MSKOR8 SF_MSKOR8B "s_waitcnt lgkmcnt(0)"
                           : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]),
                             "+v"(o[8]), "+v"(o[9]), "+v"(o[10]), "+v"(o[11]), "+v"(o[12]), "+v"(o[13]), "+v"(o[14]), "+v"(o[15]), [d] "+v"(dat)
                           : [m] "v"(msk), [tab] "n"(L_TABLE)
                           : "memory");
            } else {
              asm volatile(SF_MSKOR8 "s_waitcnt lgkmcnt(0)"
                           : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), [d] "+v"(dat)
                           : [m] "v"(msk), [tab] "n"(L_TABLE)
                           : "memory");
            }
#undef SF_MSKOR8B
#undef SF_MSKOR8
#undef SF_MSKOR
            if constexpr (STRIDE2) {
              if (!(lane & 1u)) {
#pragma unroll
                for (uint32_t sl = 0; sl < NSL; ++sl) s_ans[sl * 32 + (lane >> 1)] = (uint16_t)o[sl];
              }
            } else {
#pragma unroll
              for (uint32_t sl = 0; sl < NSL; ++sl) s_ans[sl * 64 + lane] = (uint16_t)o[sl];
            }
            __builtin_amdgcn_s_setprio(2);
          }
          lds_barrier();  // the step is in the table, the answers are in place
        }
        // STRIDE2: the last odd position of the last step that ran has no successor in its step: no match (see below)
        if (STRIDE2 && t == 0 && nsteps < kRound / STEP) smem[L_LEN4 + nsteps * (STEP / 2)] = 0;
        if (t < kHistStride) s_hist[t] = hsave;  // (the last barrier above follows the last read of a post)
        if (r + 1 < nrounds) {  // the next round's bytes
          pre_lo = load4(rb + kRound + kLook + 8 * t);
          pre_hi = load4(rb + kRound + kLook + 8 * t + 4);
        }
      } else {
        // @phase match.setup2 trips=1
        // A step has 512 searches for 1024 threads: the even positions of 1024 (STRIDE2), or all of 512 (thorough).  The
        // two halves of the workgroup (waves 0..7 and 8..15) take the steps in turn and split a search in two: in the
        // INTERVAL before step `it` is inserted, the half whose turn it is does the first part of step `it` -- bytes,
        // hash, the far levels as the table holds them before the step, their ranks -- while the other half does the
        // second part of step it - 1, whose first part it did an interval ago -- the near candidate as the table stands
        // after that step's insertions, the winner, its extension, the results.  The half that did the first part
        // inserts the step: its position and, with STRIDE2, the odd position behind it (whose bytes it has in registers
        // anyway).  Every thread is busy in every interval; a search costs one thread two.
        const uint32_t grp = wave >> 3;                   // (uniform) this wave's half
        const uint32_t tp = t & 511u;                     // index in the half
        const uint32_t ps = STRIDE2 ? 2 * tp : tp;        // the thread's searched position within a step
        const uint32_t to_rend = kRegion - (ps & (kRegion - 1));  // ... and how far its parse region's end is
        static_assert(STEP % kRegion == 0, "steps are whole regions");
        // an odd position takes over its successor's match only inside the step and the parse region
        [[maybe_unused]] const bool inh_here = tp != 0 && (tp & (kRegion / 2 - 1)) != 0;
        [[maybe_unused]] const uint32_t first_ad = kWindow - rb;  // (uniform) LDS address of the strip's first byte (no address at all after the first rounds)
        // Addresses inside a step are a uniform part (the step's offset in the round) plus what never changes for the
        // thread; constants ride in the LDS instructions' offset fields.  psw: the dword the position starts in (the
        // odd position behind it starts in the same one), sh0: where in it; cv: the low bits of the position's step code
        const uint32_t sh0 = ps & 3u, cv = STEP - 1u - ps;
        const uint32_t pswK = kWindow + (ps & ~3u);   // LDS address of that dword in step 0
        const uint32_t psK1 = kWindow - 1u + ps;      // LDS address of the byte before the position in step 0
        const uint32_t mlen0 = to_rend < kCap ? to_rend : kCap;  // bytes a match may take from the position, the input's end aside
        auto lds32 = [&](uint32_t at, uint32_t off) { return *reinterpret_cast<const uint32_t*>(smem + at + off); };
        // (uniform) the round's staging slots.  (Addressed as base + the position's offset in the round: with the
        // constant folded into the base -- an odd base below the array, odd offsets -- the same stores cost 1 GiB of
        // extra HBM reads per GiB of input: FETCH_SIZE 274 against 145 MiB per 256 MiB, tools/exp/pmc_fetch_variants.sh)
        uint8_t* const gst = reinterpret_cast<uint8_t*>(gi) + stage_off;
        // first part -> second part.  (LONG: f_m0.. f_q1 carry the four far candidates as KEYS rank << 16 | 0xFFFF - distance)
        // A bucket is named by its BYTE OFFSET in its table (hash << 2), made by one shift and one mask of the product: the
        // index would be shifted left again for every read and every atomic (`v_lshlrev_b32`: 4.1 cycles against 2.3 for
        // the mask; five of them per search in the phase where an instruction costs most, section 3 K1 "Round 6")
        auto bucket_of = [](uint32_t product) { return (product >> (30 - HB)) & (((1u << HB) - 1u) << 2); };
        static_assert(HB <= kHashBits && 30 - HB > 0, "bucket offsets");
        auto tab_at = [&](uint32_t* table, uint32_t off) -> uint32_t& { return *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(table) + off); };
        uint32_t f_a0 = 0, f_a1 = 0, f_h = 0, f_m0 = 0, f_m1 = 0, f_q0 = 0, f_q1 = 0, f_maxlen = 0;
        // @phase match.loop trips=9 depth=2 note=nine intervals for eight steps
        for (uint32_t it = 0; it <= nsteps; ++it) {
          uint32_t ins_h = 0, ins_v = 0;
          [[maybe_unused]] uint32_t ins2_h = 0, ins2_v = 0;  // LONG: the same for the seven-byte table; STRIDE2: for the odd position behind
          bool has_ins = false;                           // (uniform) this wave inserts in this interval
          const uint32_t code_it = (rb / STEP + it - ebase + 1) << SH;  // (uniform) step code of step `it`
          // ... | the position's low bits: ONE v_or (the compiler would derive it from an address it has at hand, in three)
          auto code_of = [&](uint32_t low) {
            uint32_t c;
            asm("v_or_b32 %0, %1, %2" : "=v"(c) : "s"(code_it), "v"(low));
            return c;
          };
          if ((it & 1) == grp) {
            if (it < nsteps) {
              // @phase match.first trips=4 depth=2 note=a wave does the first part of every second step
              // ---- first part of the search at position ps of step `it` ----
              const uint32_t sb = it * STEP;            // (uniform) the step's first position, round-relative
              const uint32_t wb = sb + pswK;
              const uint32_t d0 = lds32(wb, 0), d1 = lds32(wb, 4), d2 = lds32(wb, 8);
              const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, sh0), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh0);
              const uint32_t hmul = a0 * 2654435761u;
              const uint32_t h = bucket_of(hmul);       // (byte offset of the bucket: see bucket_of)
              const uint32_t farv = tab_at(s_table, h);
              if constexpr (STRIDE2) {
                // the odd position behind this one is only inserted: its four bytes are in the registers already, its
                // bucket is read beside this one's
                static_assert(!STRIDE2 || HB == kHashBits, "the odd position's bucket is in the one table");
                ins2_h = bucket_of(__builtin_amdgcn_alignbyte(a1, a0, 1) * 2654435761u);
                ins2_v = tab_at(s_table, ins2_h);
              }
              [[maybe_unused]] uint32_t h2 = 0, farv2 = 0;
              if constexpr (LONG) {
                // bytes 4..6 join the hash (the specification's long hash: two multiplicative hashes xored)
                h2 = bucket_of(hmul ^ ((a1 & 0xFFFFFFu) * 0x85EBCA6Bu));
                farv2 = tab_at(s_table2, h2);
              }
              const uint32_t f0 = farv >> 16, f1 = farv & 0xFFFFu;
              const uint32_t m0 = entry_rel<SH>(f0), m1 = entry_rel<SH>(f1);  // coded positions + 1, from the epoch's start
              const uint32_t c0 = m0 + (K - 1), c1 = m1 + (K - 1);
              // An empty or outdated entry decodes to some address that is not a candidate: it is still read (kept inside
              // the LDS allocation by a 16-bit mask that leaves real candidates alone), its rank is dropped.
              // ad - c <= kWindow  <=>  m >= thr, as signed numbers: thr >= 1 (see ebase), m < 0 for what ageing left of a
              // position in the step before the epoch
              static_assert(kWindow + kRound + kLook <= 0x10000 && 0x10000 + 16 <= K1_LDS, "candidate reads stay in LDS");
              const int32_t thr = (int32_t)((sb - (K - 1)) + ps);
              const bool ok0 = (int32_t)m0 >= thr, ok1 = DEPTH2 && (int32_t)m1 >= thr;
              // bytes a match may take from here: inside the round's valid part, the parse region and kCap (0 beyond qn).
              // (A step is a whole number of regions: the distance to the region's end does not depend on the step.)
              const uint32_t maxlen = (uint32_t)max(min((int)((qn - sb) - ps), (int)mlen0), 0);
              // candidates are ranked by their first kRank bytes; only the winner is compared to kCap
              uint32_t l0, l1 = 0;
              if constexpr (DEPTH2) rank8x2(s_data, a0, a1, c0, c1, maxlen, l0, l1);
              else l0 = rank8(s_data, a0, a1, c0, maxlen);
              f_a0 = a0; f_a1 = a1; f_h = h; f_maxlen = maxlen;
              if constexpr (LONG) {
                const uint32_t g0 = farv2 >> 16, g1 = farv2 & 0xFFFFu;
                const uint32_t m2 = entry_rel<SH>(g0), m3 = entry_rel<SH>(g1);
                const uint32_t c2 = m2 + (K - 1), c3 = m3 + (K - 1);
                // a position without kLongBytes bytes left in the strip reads nothing from the seven-byte table
                const bool lng = rb + sb + ps + kLongBytes <= n;
                const bool ok2 = lng && (int32_t)m2 >= thr, ok3 = lng && (int32_t)m3 >= thr;
                uint32_t l2, l3;
                rank8x2(s_data, a0, a1, c2, c3, maxlen, l2, l3);
                const uint32_t ad = kWindow + sb + ps;
                auto key = [&](bool ok, uint32_t l, uint32_t c) { return ok ? (l << 16) | (0xFFFFu - (ad - c)) : 0u; };
                f_m0 = key(ok0, l0, c0); f_m1 = key(ok1, l1, c1); f_q0 = key(ok2, l2, c2); f_q1 = key(ok3, l3, c3);
                ins2_h = h2;
                ins2_v = __builtin_amdgcn_alignbit(code_of(cv), farv2, 16);
              } else {
                f_q0 = c0; f_q1 = c1;
                f_m0 = ok0 ? l0 : 0u;
                f_m1 = ok1 ? l1 : 0u;
              }
              ins_h = h;
              ins_v = __builtin_amdgcn_alignbit(code_of(cv), farv, 16);
              if constexpr (STRIDE2) ins2_v = __builtin_amdgcn_alignbit(code_of(cv - 1u), ins2_v, 16);
              has_ins = true;
            }
          } else {
            if (it >= 1) {
              // @phase match.second trips=4 depth=2
              // ---- second part of the search at position ps of step it - 1 ----
              const uint32_t sb = (it - 1) * STEP;      // (uniform) that step's first position, round-relative
              const uint32_t ad1 = sb + psK1;           // LDS address of the byte before the position
              const uint32_t wb = sb + pswK;
              const uint32_t a0 = f_a0, a1 = f_a1, maxlen = f_maxlen;
              uint32_t neare = NEAR ? tab_at(s_table, f_h) : 0u;
              uint32_t pbyte = STRIDE2 ? smem[ad1] : 0u;  // the odd position's byte (used if a match is found)
              // this position's bytes 8..15 (for the winner's extension, if it comes to that)
              uint32_t d2 = lds32(wb, 8), d3 = lds32(wb, 12), d4 = lds32(wb, 16);
              asm volatile("" : "+v"(neare), "+v"(pbyte), "+v"(d2), "+v"(d3), "+v"(d4));  // one round trip for all
              // longest wins; ties go to the smaller distance: near, then the newer far level
              uint32_t best = 0, bq = ad1 + 1u;
              if constexpr (NEAR) {
                // the step's first position with this hash: this one's own entry at the latest, so the bucket is not empty
                // (and it is of this very step: only its index in the step has to be decoded)
                uint32_t nc;
                const uint32_t nlow = (neare >> 16) & (STEP - 1u);
                asm("v_sub_u32 %0, %1, %2" : "=v"(nc) : "s"(kWindow + sb + STEP - 1u), "v"(nlow));
                const bool okn = nlow > cv;  // nc < ad (nc == ad: this position is the first, it has no near candidate)
                const uint32_t ln = rank8(s_data, a0, a1, nc, maxlen);
                best = okn ? ln : 0u;
                bq = nc;
              }
              if constexpr (LONG) {
                // the largest key: the longest rank, the nearest of those
                const uint32_t ad = ad1 + 1u;
                const uint32_t kn = best ? (best << 16) | (0xFFFFu - (ad - bq)) : 0u;
                const uint32_t kb = max(max(max(f_m0, f_m1), max(f_q0, f_q1)), kn);
                best = kb >> 16;
                bq = best ? ad - (0xFFFFu - (kb & 0xFFFFu)) : ad;
              } else {
                if (f_m0 > best) { best = f_m0; bq = f_q0; }
                if (f_m1 > best) { best = f_m1; bq = f_q1; }
              }
              const uint32_t bd1 = ad1 - bq;  // distance - 1 (what is staged; nothing reads it where there is no match)
              const uint32_t cbyte = STRIDE2 ? s_bytes[(bq - 1) & 0xFFFFu] : 0u;  // the byte in front of the winner (any bq reads inside LDS)
              if (best == kRank) {
                // the winner's next eight bytes, only where its first kRank all matched
                const uint32_t a2 = __builtin_amdgcn_alignbyte(d3, d2, sh0), a3 = __builtin_amdgcn_alignbyte(d4, d3, sh0);
                const uint32_t lx = kRank + cmp8<kRank / 4>(s_data, a2, a3, bq);  // maxlen <= kCap does the capping
                best = lx < maxlen ? lx : maxlen;
              }
              // a 4-byte match farther than kFar4 costs more bits than four literals: drop it.  (maxlen <= n - p, so a
              // position without kMinMatch bytes left cannot reach kMinMatch.)
              const bool ok = best >= (bd1 >= kFar4 ? kMinMatch + 1 : kMinMatch);
              const uint32_t len4 = ok ? best - 3 : 0u;
              if constexpr (STRIDE2) {
                // the odd position in front: the same match one byte longer if its byte fits too (and the candidate is
                // not the strip's first byte: the copy would start before the strip)
                const bool inh = ok && inh_here && pbyte == cbyte && bq != first_ad;
                const uint32_t len4o = inh ? (best < kCap ? best - 2 : kCap - 3) : 0u;
                // one staging slot per EVEN position (an odd position that has a match has its successor's distance)
                *reinterpret_cast<uint16_t*>(gst + (ad1 - (kWindow - 1u))) = (uint16_t)bd1;  // only read where the length says there is a match
                // 4-bit lengths, SHIFTED by one position: byte j = {position 2j - 1, position 2j} of the round -- the pair
                // this thread knows
                smem[L_LEN4 + ((sb >> 1) + tp)] = (uint8_t)(len4o | (len4 << 4));
              } else {
                *reinterpret_cast<uint16_t*>(gst + 2u * (ad1 - (kWindow - 1u))) = (uint16_t)bd1;
                // two lanes' 4-bit lengths -> one byte (the odd lane's value comes over the DPP network), stored by the
                // even lanes.  All 64 lanes of the wave are active here, so the execution mask is switched and restored by
                // hand: two scalar moves instead of the save / branch / restore a divergent `if` compiles to (smem sits at
                // LDS address 0: it is the kernel's only __shared__ object)
                uint32_t v = len4;
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true) << 4;
                // (the store is invisible to the compiler's counter tracking: waited for here, ahead of the barrier behind it)
                asm volatile("s_mov_b64 exec, %2\n\tds_write_b8 %0, %1 offset:%3\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                             :: "v"((sb + tp) >> 1), "v"(v), "s"(0x5555555555555555ull), "n"(L_LEN4) : "memory");
              }
            }
          }
          if constexpr (SF_PROBE_MATCH > 0) { uint32_t pr = t; probe_valu<SF_PROBE_MATCH>(pr); }
          // @phase match.barrier trips=8 depth=2
          if (it == nsteps) break;
          lds_barrier();  // every read of the table as it stands before step `it` precedes the step's insertions
          // @phase match.insert trips=4 depth=2
          if (has_ins) {  // (the half that did the step's first part)
            // {code, old newest}: the upper half of code:bucket.  (Positions without kMinMatch bytes left insert like the
            // rest, which nothing can observe: every position after them in the strip is such a position too and takes no
            // match, the next strip starts from an empty table.)  A lane whose predecessor in the wave has the same bucket
            // need not insert: that one's position is lower, its code larger (a run would otherwise serialise the wave's
            // atomics)
            const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp((int)~ins_h, (int)ins_h, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            if (hp != ins_h) atomicMax(&tab_at(s_table, ins_h), ins_v);
            if constexpr (LONG) {
              const uint32_t hp2 = (uint32_t)__builtin_amdgcn_update_dpp((int)~ins2_h, (int)ins2_h, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
              if (hp2 != ins2_h) atomicMax(&tab_at(s_table2, ins2_h), ins2_v);
            }
            if constexpr (STRIDE2) {
              // the odd position behind: the same, among the wave's odd positions (and its own even position's bucket has
              // the larger code already)
              const uint32_t hp2 = (uint32_t)__builtin_amdgcn_update_dpp((int)~ins2_h, (int)ins2_h, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
              if (hp2 != ins2_h && ins2_h != ins_h) atomicMax(&tab_at(s_table, ins2_h), ins2_v);
            }
          }
          // @phase match.barrier2 trips=8 depth=2
          lds_barrier();  // insertions complete before the near reads
        }
        // @phase match.tail trips=1
        // STRIDE2: the last position of the last step that ran is odd and has no successor in its step: no match.  Its length
        // lives in the low half of the byte behind the step's (nobody wrote it in this round; after all steps it is the pad)
        if (STRIDE2 && t == 0 && nsteps < kRound / STEP) smem[L_LEN4 + nsteps * (STEP / 2)] = 0;
      }
      // @phase parse.begin trips=1 note=barrier, staged-distance load
      __builtin_amdgcn_s_setprio(1);  // the parse: behind the other workgroup's match, ahead of its emit
      __syncthreads();  // (also: the staged distances are visible to the whole workgroup)
      stamp(1);
      // the lane's eight staged distances: asked for now, used by the extensions and the emit.  Every wave has them
      // before any wave writes an item (the barrier between walk and emit waits for outstanding loads)
      // (the load is only waited for behind the take pass and the transfer functions, which do not need it)
      uint32_t D0 = 0, D1 = 0, D2 = 0, D3 = 0;
      uint4 Dld = make_uint4(0, 0, 0, 0);
      if constexpr (STRIDE2) {
        // four slots: the lane's even positions; position 7's successor is the next lane's position 0
        const uint2 D = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(gi) + (uint64_t)(stage_off + 8u * t));
        Dld.x = D.x; Dld.y = D.y;
      } else {
        Dld = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(gi) + (uint64_t)(stage_off + 16u * t));
      }
      // @phase inherit
      auto staged_arrive = [&]() {
        D0 = Dld.x; D1 = Dld.y;
        if constexpr (STRIDE2) {
          asm volatile("" : "+v"(D0), "+v"(D1));
          D2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)D0, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);  // (lane 63: its position 7 ends the region, never inherits)
        } else {
          D2 = Dld.z; D3 = Dld.w;
          asm volatile("" : "+v"(D0), "+v"(D1), "+v"(D2), "+v"(D3));  // four registers, not an indexable vector (scratch)
        }
      };
      auto dm1_of = [&](uint32_t k) {  // distance - 1 of the lane's position k (0..7)
        if constexpr (STRIDE2) {
          const uint32_t e = (k + 1u) >> 1;            // even position 2e holds it (an odd k: its successor's); e = 0..4
          const uint32_t w = (e & 4u) ? D2 : ((e & 2u) ? D1 : D0);
          return (e & 1u) ? w >> 16 : w & 0xFFFFu;
        } else {
          const uint32_t lo = (k & 2u) ? D1 : D0, hi = (k & 2u) ? D3 : D2;
          const uint32_t w = (k & 4u) ? hi : lo;
          return (k & 1u) ? w >> 16 : w & 0xFFFFu;
        }
      };

      // @phase parse.take trips=1
      // ---- parse: wave-local.  Thread t owns the eight positions [8t, 8t+8) of the round, a wave one
      // kRegion-byte parse region (matches never cross it), so the greedy/lazy chain of a region is
      // resolved inside one wave, in registers, with no barrier:
      //   take    : from the lane's dword of 4-bit lengths (+ the next lane's for the look-ahead) the
      //             positions whose match the greedy/lazy rule would take
      //   walk    : every lane follows the chain through its eight positions from an entry offset
      //             (speculatively 0) and reports where the chain leaves it; a lane's true entry is
      //             the largest exit of the lanes before it (exclusive prefix maximum over the wave;
      //             lanes the chain jumps over report nothing).  Lanes whose entry changed walk again,
      //             until nothing changes -- lane k is final after k rounds at the latest, usually
      //             after two or three, since chains that meet stay together.
      //   count   : tokens / matches per lane, wave scan, one total per wave to LDS
      // then one barrier and
      //   emit    : every lane writes its (at most eight) items at their compact place, + histogram
      const uint32_t pb = 8 * t;                       // the lane's first position, round-relative
      if constexpr (SF_PROBE_PARSE > 0) { uint32_t pr = t; probe_valu<SF_PROBE_PARSE>(pr); }
      const uint32_t nv = qn > pb ? (qn - pb < 8 ? qn - pb : 8u) : 0u;  // its valid positions
      uint32_t T = 0;                                  // take bits of the eight positions
      uint32_t N = 0;                                  // their 4-bit lengths
      uint32_t c0 = 0, c1 = 0, k0 = 0, k1 = 0;         // a byte per position: 4-bit length (0..13) | 0x80 where T says take
      // (uniform per wave) behind a chunk that took the stored fast path only the first kSkipProbe positions of the span were
      // searched: the waves behind them have nothing to parse -- every position a literal
      const bool lit_wave = short_probe && rc == 0 && wave >= kSkipProbe / kRegion;
      if (!lit_wave) {
        // lazy deferral looks up to `lazy` positions ahead, inside the region and the input
        uint64_t W;
        if constexpr (STRIDE2) {
          // the lengths are stored shifted by one position (see the match phase): dword t holds positions 8t-1 .. 8t+6
          const uint32_t w0 = s_len4[t], w1 = s_len4[t + 1];
          W = ((uint64_t)w0 | ((uint64_t)(lane == 63 ? w1 & 15u : w1) << 32)) >> 4;
        } else {
          const uint32_t wn = lane == 63 ? 0u : s_len4[t + 1];
          W = (uint64_t)s_len4[t] | ((uint64_t)wn << 32);
        }
        // positions at or beyond qn carry stale lengths of an earlier round: clear them.  (Only a strip's last round can be
        // short -- uniform; in a full round the one dword that reaches past the round belongs to a lane 63, whose high half
        // is masked above.)
        if (qn < kRound) {
          const uint32_t left = qn > pb ? qn - pb : 0u;  // valid positions from this dword's first on
          if (left < 16) W &= left ? ((1ull << (4 * left)) - 1ull) : 0ull;
        }
        N = (uint32_t)W;
        // All eight positions at once, a byte per position (SWAR): the 4-bit lengths of positions 0..10 are spread into
        // the bytes of three dwords; b > a + j for bytes a, b <= 15 is bit 7 of (b + 0x7F - j) - a, and no byte borrows from
        // its neighbour (every byte stays in 112..142); the eight bit-7s are gathered by two byte dot products.
        const uint32_t wl = (uint32_t)W, wh = (uint32_t)(W >> 32);
        const uint32_t le = wl & 0x0F0F0F0Fu, lo4 = (wl >> 4) & 0x0F0F0F0Fu;   // positions 0,2,4,6 | 1,3,5,7
        const uint32_t he = wh & 0x0F0F0F0Fu, ho4 = (wh >> 4) & 0x0F0F0F0Fu;   // positions 8,10,.. | 9,11,..
        c0 = __builtin_amdgcn_perm(lo4, le, 0x05010400u);                      // bytes = positions 0,1,2,3
        c1 = __builtin_amdgcn_perm(lo4, le, 0x07030602u);                      // positions 4..7
        const uint32_t c2 = __builtin_amdgcn_perm(ho4, he, 0x05010400u);       // positions 8..11
        // uniform: which look-aheads count (lazy = 0..3)
        const uint32_t m1 = lazy >= 1 ? 0x80808080u : 0u, m2 = lazy >= 2 ? 0x80808080u : 0u, m3 = lazy >= 3 ? 0x80808080u : 0u;
        auto take4 = [&](uint32_t cur, uint32_t nx) {  // cur: four positions, nx: the four behind them
          const uint32_t a1 = __builtin_amdgcn_alignbyte(nx, cur, 1), a2 = __builtin_amdgcn_alignbyte(nx, cur, 2),
                         a3 = __builtin_amdgcn_alignbyte(nx, cur, 3);
          const uint32_t t1 = (a1 + 0x7F7F7F7Fu) - cur, t2 = (a2 + 0x7E7E7E7Eu) - cur, t3 = (a3 + 0x7D7D7D7Du) - cur;
          const uint32_t defer = (t1 & m1) | (t2 & m2) | (t3 & m3);
          return (cur + 0x7F7F7F7Fu) & ~defer & 0x80808080u;  // bit 7: a match, and nothing ahead says wait
        };
        k0 = take4(c0, c1);
        k1 = take4(c1, c2);
        // 0x80 * (b0 + 2 b1 + 4 b2 + 8 b3) + 0x80 * (16 b4 + ...): bits 7..14
        const uint32_t bits = __builtin_amdgcn_udot4(k1, 0x80402010u, __builtin_amdgcn_udot4(k0, 0x08040201u, 0u, false), false) >> 7;
        T = bits;
      }
      stamp(2);
      // @phase parse.transfer trips=1
      uint32_t marks = 0;                              // chain positions among the eight
      uint32_t cap_mp = 8, cap_len = 0;                // the capped chain match that was extended (at most one per lane)
      bool cap_run = false;                            // ... and whether it turned out to be a run (distance 1)
      uint32_t exit_abs = 0;                           // region-relative position where the chain leaves this lane (0: not on it)
      if (lit_wave) {
        marks = (1u << nv) - 1u;
      } else {
        const uint32_t lb = 8 * lane;                  // the lane's first position, region-relative
        const uint32_t rend = (pb & ~(kRegion - 1)) + kRegion < qn ? (pb & ~(kRegion - 1)) + kRegion : qn;  // region end, round-relative
        // The lane's TRANSFER FUNCTION, for all eight possible entries at once (a byte per entry, SWAR): where the chain
        // leaves the lane when it enters at position e (ex: 8..23 = the position behind the lane's eight it jumps to,
        // 0x80 | k = it takes the match at k that was capped at match time, whose length only the extension knows)
        // and which of the lane's positions it visits (mk).  A position's successor is position + 1 (literal) or
        // + length (taken match); successors of successors by pointer doubling -- v_perm_b32 IS an eight-entry byte
        // table -- three times, since a chain has at most eight hops inside a lane.  The reconcile rounds below then
        // look an entry up instead of walking it.
        uint32_t elo, ehi, mlo = 0x08040201u, mhi = 0x80402010u;
        {
          auto spread = [](uint32_t bit7s) { return (bit7s >> 7) * 0xFFu; };  // 0x80 -> 0xFF in every byte
          const uint32_t t0 = spread(k0), t1 = spread(k1);
          const uint32_t s0 = (t0 & (c0 + 0x03030303u)) | (~t0 & 0x01010101u), s1 = (t1 & (c1 + 0x03030303u)) | (~t1 & 0x01010101u);
          constexpr uint32_t kCap4 = (kCap - 3) * 0x01010101u;
          const uint32_t cp0 = spread(~((c0 ^ kCap4) + 0x7F7F7F7Fu) & k0), cp1 = spread(~((c1 ^ kCap4) + 0x7F7F7F7Fu) & k1);
          elo = (cp0 & 0x83828180u) | (~cp0 & (s0 + 0x03020100u));
          ehi = (cp1 & 0x87868584u) | (~cp1 & (s1 + 0x07060504u));
#pragma unroll
          for (uint32_t j = 0; j < 3; ++j) {
            const uint32_t tm0 = spread((elo + 0x78787878u) & 0x80808080u), tm1 = spread((ehi + 0x78787878u) & 0x80808080u);  // final already
            const uint32_t g0 = __builtin_amdgcn_perm(ehi, elo, elo), g1 = __builtin_amdgcn_perm(ehi, elo, ehi);
            const uint32_t h0 = __builtin_amdgcn_perm(mhi, mlo, elo), h1 = __builtin_amdgcn_perm(mhi, mlo, ehi);
            mlo |= h0 & ~tm0;
            mhi |= h1 & ~tm1;
            elo = (tm0 & elo) | (~tm0 & g0);
            ehi = (tm1 & ehi) | (~tm1 & g1);
          }
        }
        // @phase inherit
        const uint32_t vmask = (1u << nv) - 1u;          // the lane's valid positions
        // The chain enters at local offset e (< nv).  A match that was capped at match time (kCap bytes or more: it leaves
        // the lane whatever its length) is extended here, once per position: the lane itself looks at the next kLaneExt
        // bytes, which settles nearly every such match of ordinary data; one that is still going then (a run, a repeated
        // record) is noted for the wave
        constexpr uint32_t kLaneExt = 16;
        // cap_mp: 0..7 the position whose extended length cap_len holds, 8 none, 16 + p: position p waits for the wave
        auto walk = [&](uint32_t e) {
          cap_mp = cap_mp >= 16 ? 8u : cap_mp;           // a request of an earlier walk is void
          const uint32_t ex = __builtin_amdgcn_perm(ehi, elo, e) & 0xFFu;
          uint32_t pos = ex;
          // @phase +ext trips=0.5 note=some lane enters at a capped match
          if (ex & 0x80u) {                              // capped at match time: extend (once per position)
            const uint32_t mp = ex & 7u;
            uint32_t len;
            if (cap_mp != mp) {
              cap_run = false;
              const uint32_t xpa = kWindow + pb + mp, xca = xpa - 1u - dm1_of(mp);
              const uint32_t xmax = rend - (pb + mp) < 258u ? rend - (pb + mp) : 258u;
              // the next kLaneExt bytes of both strings, straight-line (bytes past xmax are cut off below)
              static_assert(kLaneExt == 16, "four dwords per string");
              const uint32_t ia = xpa + kCap, ja = xca + kCap;
              const uint32_t* ip = s_data + (ia >> 2);
              const uint32_t* jp = s_data + (ja >> 2);
              const uint32_t i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3], i4 = ip[4];
              const uint32_t j0 = jp[0], j1 = jp[1], j2 = jp[2], j3 = jp[3], j4 = jp[4];
              const uint32_t si = ia & 3, sj = ja & 3;
              const uint32_t x0 = __builtin_amdgcn_alignbyte(i1, i0, si) ^ __builtin_amdgcn_alignbyte(j1, j0, sj);
              const uint32_t x1 = __builtin_amdgcn_alignbyte(i2, i1, si) ^ __builtin_amdgcn_alignbyte(j2, j1, sj);
              const uint32_t x2 = __builtin_amdgcn_alignbyte(i3, i2, si) ^ __builtin_amdgcn_alignbyte(j3, j2, sj);
              const uint32_t x3 = __builtin_amdgcn_alignbyte(i4, i3, si) ^ __builtin_amdgcn_alignbyte(j4, j3, sj);
              // first differing bit of x3:x2:x1:x0 (all ones when there is none: ffbl(0) = 0xFFFFFFFF survives the ORs)
              const uint32_t fb = min(min(ffbl(x0), ffbl(x1) | 32u), min(ffbl(x2) | 64u, ffbl(x3) | 96u));
              const bool open = fb >= 8 * kLaneExt;      // all kLaneExt bytes equal
              uint32_t l = kCap + (open ? kLaneExt : fb >> 3);
              l = l < xmax ? l : xmax;
              if (open && l < xmax) {
                cap_mp = 16 + mp;                        // still equal after kLaneExt more bytes: the wave's turn
                len = kCap + kLaneExt;                   // provisional (the chain has left the lane anyway)
              } else {
                cap_mp = mp;
                cap_len = l;
                len = l;
              }
            } else {
              len = cap_len;
            }
            pos = mp + len;
          }
          // @phase inherit
          marks = __builtin_amdgcn_perm(mhi, mlo, e) & vmask;
          exit_abs = lb + pos;                           // (pos >= 8: the chain has left the lane)
        };
        // A long capped match, by the whole wave: lane l compares the four bytes at offset kCap + kLaneExt + 4 l of
        // the two strings, so one pass covers the 258 bytes a match can have.  Only the FIRST waiting lane is served
        // per reconcile round: the match usually jumps over the lanes behind it, whose speculative requests then
        // never have to be looked at (on a run of zeros every lane has one)
        static_assert(kCap + kLaneExt + 4 * 64 >= 258, "one pass of the wave covers the longest match");
        auto extend_first = [&]() {
          const uint64_t need = __builtin_amdgcn_ballot_w64(cap_mp >= 16);
          if (need == 0) return false;
          // @phase +serve trips=0.1 note=a lane's match is still open after 32 bytes: the wave extends it
          const uint32_t src_lane = (uint32_t)__builtin_ctzll(need);
          const uint32_t xat = (uint32_t)__builtin_amdgcn_readlane((int)(pb + cap_mp - 16), (int)src_lane);  // round-relative position
          const uint32_t xd = (uint32_t)__builtin_amdgcn_readlane((int)dm1_of((cap_mp - 16) & 7), (int)src_lane) + 1u;
          const uint32_t xmax = rend - xat < 258u ? rend - xat : 258u;  // (rend is the same for the whole wave)
          const uint32_t ia = kWindow + xat + kCap + kLaneExt + 4 * lane, ja = ia - xd;
          const uint32_t i0 = s_data[ia >> 2], i1 = s_data[(ia >> 2) + 1];
          const uint32_t j0 = s_data[ja >> 2], j1 = s_data[(ja >> 2) + 1];
          const uint32_t x = __builtin_amdgcn_alignbyte(i1, i0, ia & 3) ^ __builtin_amdgcn_alignbyte(j1, j0, ja & 3);
          const uint64_t diff = __builtin_amdgcn_ballot_w64(x != 0);
          uint32_t l = kCap + kLaneExt + 4 * 64;       // no difference within reach
          if (diff) {
            const uint32_t dl = (uint32_t)__builtin_ctzll(diff);
            const uint32_t xx = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)dl);
            l = kCap + kLaneExt + 4 * dl + ((uint32_t)__builtin_ctz(xx) >> 3);
          }
          l = l < xmax ? l : xmax;
          // A run of one byte value: when the match's bytes all equal the byte before it, it is coded at distance 1 (an
          // overlapping copy, /root/reference/src/decompress.cpp:388-398) with the run's length.  Lane k compares the
          // eight bytes from offset 8k with the eight before them shifted by one: 33 lanes cover the longest match
          static_assert(33 * 8 >= 258, "the run check covers the longest match");
          bool run = false;
          {
            const uint32_t q1 = kWindow + xat + 8 * lane - 1;  // the byte before the lane's eight
            const uint32_t* ep = s_data + (q1 >> 2);
            const uint32_t e0 = ep[0], e1 = ep[1], e2 = ep[2], e3 = ep[3];
            const uint32_t sh = q1 & 3;
            const uint32_t w0 = __builtin_amdgcn_alignbyte(e1, e0, sh), w1 = __builtin_amdgcn_alignbyte(e2, e1, sh),
                           w2 = __builtin_amdgcn_alignbyte(e3, e2, sh);
            const uint32_t y0 = w0 ^ __builtin_amdgcn_alignbyte(w1, w0, 1), y1 = w1 ^ __builtin_amdgcn_alignbyte(w2, w1, 1);
            const uint64_t rdiff = __builtin_amdgcn_ballot_w64((y0 | y1) != 0 && lane < 33);
            uint32_t rl = 33 * 8;
            if (rdiff) {
              const uint32_t dl = (uint32_t)__builtin_ctzll(rdiff);
              const uint32_t f = first_bit64((uint32_t)__builtin_amdgcn_readlane((int)y0, (int)dl),
                                             (uint32_t)__builtin_amdgcn_readlane((int)y1, (int)dl));
              rl = 8 * dl + (f >> 3);
            }
            rl = rl < xmax ? rl : xmax;
            if (rl >= l && rb + xat >= 1) { run = true; l = rl; }  // (the strip's first byte has none before it)
          }
          if (lane == src_lane) {
            cap_run = run;  // the emit then writes distance 1 for this match
            cap_mp -= 16;
            cap_len = l;
            exit_abs = lb + cap_mp + l;                // the walk had left the lane at this match
          }
          return true;
        };
        // @phase parse.walk0 trips=1 note=speculative walk from entry 0
        staged_arrive();
        uint32_t entry = 0;
        if (nv) walk(0);
        // @phase parse.reconcile trips=3.5 depth=2 note=2.5 rounds on text + the round that finds nothing changed
#pragma unroll 1
        for (uint32_t round = 0; round < 128; ++round) {
          const bool served = extend_first();
          const uint32_t pm = wave_excl_max(exit_abs);
          const uint32_t ne = pm > lb ? pm - lb : 0u;  // where the chain enters this lane (>= nv: it jumps over it)
          const bool changed = ne != entry && nv != 0;
          if (__builtin_amdgcn_ballot_w64(changed) == 0 && !served) break;
          if constexpr (STAMPS) st_acc[7] += 1;  // diagnostic: reconcile rounds of wave 0
          if (changed) {
            entry = ne;
            if (ne >= nv) { marks = 0; exit_abs = 0; cap_mp = cap_mp >= 16 ? 8u : cap_mp; }
            else walk(ne);
          }
        }
      }
      stamp(3);
      // @phase parse.counts trips=1
      // tokens (low half) and matches (high half: a match takes two items) before this lane, per wave, per round
      const uint32_t cm = marks & T;                   // chain positions that are matches
      const uint32_t mine = (uint32_t)__popc(marks) | ((uint32_t)__popc(cm) << 16);
      const uint32_t incl = wave_incl_add(mine);
      if (lane == 63) s_wtot[wave] = incl;
      __syncthreads();
      uint32_t wbase, rtotal;
      {
        // totals of the waves before this one, and of all: one scan over the sixteen entries
        const uint32_t w = lane < K1_WAVES ? s_wtot[lane] : 0u;
        const uint32_t wi = wave_incl_add(w);
        rtotal = (uint32_t)__builtin_amdgcn_readlane((int)wi, K1_WAVES - 1);
        wbase = (uint32_t)__builtin_amdgcn_readlane((int)(wi - w), (int)wave);
      }
      // Stored fast path, decided HERE (uniform), as soon as the probe span's token count is known: nearly all literals ->
      // the WHOLE chunk is literals, this round included (its few matches are dropped: such a chunk is stored anyway, or
      // a Huffman block of literals).  Nothing of it is written: an item IS the position's byte, and k_emit takes the
      // items from the input in the rare case that the chunk is not stored (kItemsSkipped in nitems)
      static_assert(kSkipSpan == kRound, "the probe span is the chunk's first round");
      const bool skip_now = rc == 0 && fast_skip && (n - cstart) > kSkipSpan && (rtotal & 0xFFFFu) >= kSkipSpan - kSkipSlack;
      // tokens before each 1024-byte sub-index region (two waves): where the decoder's region lanes start
      if (!skip_now && (wave & (kSubBytes / kRegion - 1)) == 0 && lane == 0)
        rtok_out[chunk * kSubRegions + rc * kRSubs + wave / (kSubBytes / kRegion)] = tot_tok + (wbase & 0xFFFFu);
      stamp(4);
      __builtin_amdgcn_s_setprio(0);  // emit, flush and the next round's stage: nothing waits for them

      // @phase skip_now trips=0
      if (__builtin_expect(skip_now, 0)) {
        // (qn == kRound here: the chunk is longer than the span.)  The REST of the chunk is taken right here, in one go: its
        // bytes are only counted, so its rounds need no window, no table and no barrier of their own; the window the next
        // chunk's probe needs comes from the input again (lds_stale, see the stage).  Schedule only: the chunk's tokens,
        // counts and sub-index are what the round-by-round version wrote.
        // Round 6, STORED BY THE PROBE (specification: probe_span_is_noise, fast_skip == 2: the block type is ours to choose):
        // a full chunk whose probe span's BYTES are as good as uniform -- their entropy, the plan's fixed point, within
        // kStoreMargin bytes of the span -- has no tokens at all, which k_plan stores; its other 24 KiB are never fetched.
        const uint32_t clen = (n - cstart) < kChunk ? (n - cstart) : kChunk;  // the chunk's bytes
        const uint2 B = *reinterpret_cast<const uint2*>(&s_bytes[kWindow + 8 * t]);
        bool pstore = false;  // (uniform)
        if (clen == kChunk) {
          // The byte counts.  On the histogram itself a wave's 64 random bytes pile up on the LDS banks (one atomic
          // instruction: six to eight cycles).  The window is dead from here on (lds_stale), so its 32 KiB hold 32 COPIES
          // of the 256 counters, copy = lane mod 32 at dword byte * 32 + copy: every lane of a half-wave has a bank of its
          // own whatever the bytes are.  Then 1024 threads fold eight copies each and four neighbours meet over the DPP
          // network: every thread of quad t / 4 holds the count of byte value t / 4.
          static_assert(256 * 32 * 4 <= kWindow && kWindow / 16 == 2 * K1_THREADS, "the copies fill the window");
          static_assert(kRoundsPerChunk == 4, "the fast path's loads");
          uint4* const z4 = reinterpret_cast<uint4*>(smem + L_DATA);
          z4[t] = make_uint4(0, 0, 0, 0);
          z4[K1_THREADS + t] = make_uint4(0, 0, 0, 0);
          __syncthreads();
          uint32_t* const rep = s_data + (lane & 31u);
          auto count8 = [&](uint32_t lo, uint32_t hi) {
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) atomicAdd(&rep[(((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xFFu) * 32u], 1u);
          };
          auto fold = [&]() -> uint32_t {
            const uint4 p = z4[2 * t], q = z4[2 * t + 1];  // thread t: copies 8 (t mod 4) .. + 8 of byte value t / 4
            uint32_t sum = (p.x + p.y) + (p.z + p.w) + (q.x + q.y) + (q.z + q.w);
            sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
            sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true);
            return sum;
          };
          count8(B.x, B.y);  // the probe span's bytes (kSkipSpan == kRound: this round's)
          __syncthreads();
          if (fast_skip > 1) {
            const uint32_t f = fold();
            uint32_t e = ((t & 3u) == 0 && f) ? f * (est_log2(kSkipSpan) - est_log2(f)) : 0u;
            e = wave_sum(e);
            if (lane == 0) s_wtot[wave] = e;  // (its last readers are behind two barriers)
            __syncthreads();
            const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum(lane < K1_WAVES ? s_wtot[lane] : 0u));
            pstore = ((ent >> 8) + 7) / 8 + kStoreMargin >= kSkipSpan;
          }
          if (!pstore) {
            // the chunk's bytes behind this round: [kRound, kRound + kLook) lie in the look-ahead, the kRound bytes from there
            // on are the thread's prefetched eight (pre_lo, pre_hi: asked for in the stage, or behind the match phase), the
            // rest comes in two loads; the last kLook bytes of the last span belong to the next chunk -- or to nobody
            uint2 w[kRoundsPerChunk - 1];
            w[0] = make_uint2(pre_lo, pre_hi);
            w[1] = *reinterpret_cast<const uint2*>(sp + cstart + 2 * kRound + kLook + 8 * t);
            w[2] = make_uint2(0, 0);
            const bool tail = 8 * t + kLook < kRound;
            if (tail) w[2] = *reinterpret_cast<const uint2*>(sp + cstart + 3 * kRound + kLook + 8 * t);
            const uint32_t look = t < kLook / 4 ? s_data[(kWindow + kRound) / 4 + t] : 0u;
            count8(w[0].x, w[0].y);
            count8(w[1].x, w[1].y);
            if (tail) count8(w[2].x, w[2].y);
            if (t < kLook / 4) {
#pragma unroll
              for (uint32_t j = 0; j < 4; ++j) atomicAdd(&rep[((look >> (8 * j)) & 0xFFu) * 32u], 1u);
            }
            __syncthreads();
            const uint32_t sum = fold();
            if ((t & 3u) == 0) s_hist[t >> 2] = sum;  // (this chunk has counted nothing else: its first round's emit was dropped)
          }
        } else {  // a strip's short last chunk: byte by byte
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) atomicAdd(&s_hist[((k < 4 ? B.x : B.y) >> (8 * (k & 3))) & 0xFFu], 1u);
          for (uint32_t pos = cstart + kRound + t; pos < cstart + clen; pos += K1_THREADS) atomicAdd(&s_hist[sp[pos]], 1u);
        }
        const uint32_t r_last = (r + kRoundsPerChunk - 1 < nrounds ? r + kRoundsPerChunk - 1 : nrounds - 1);  // the chunk's last round
        const uint32_t ctok = pstore ? 0u : clen;  // tokens = items = positions -- or none
        if (t < kSubRegions) rtok_out[chunk * kSubRegions + t] = t * kSubBytes < ctok ? t * kSubBytes : ctok;
        skip = true;
        lds_stale = true;
        stale_span = short_probe ? kSkipProbe : kSkipSpan;  // (the positions of this chunk the match phase went over)
        chunk_done = true;
        tot_tok = ctok;
        tot_items = ctok;
        rtotal = 0;
        r = r_last;
      } else
      // @phase emit.head trips=1
      // ---- emit: the lane's chain positions -> items (compact, chunk order) + histogram ----
      if (marks) {
        if constexpr (SF_PROBE_EMIT > 0) { uint32_t pr = t; probe_valu<SF_PROBE_EMIT>(pr); }
        const uint32_t before = wbase + incl - mine;
        const uint32_t ib0 = 2u * (tot_items + (before & 0xFFFFu) + (before >> 16));  // byte offset of the lane's first item
        // items of the lane before position k: one per token before it, one more per match before it -- one
        // population count over both masks side by side
        const uint32_t both = marks | (cm << 8);
        auto item_at = [&](uint32_t below) { return ib0 + 2u * (uint32_t)__popc(both & (below | (below << 8))); };
        const uint2 B = *reinterpret_cast<const uint2*>(&s_bytes[kWindow + pb]);
        // first token of a sub-index region: its position is the region's first, which is always a token start,
        // so the flag can only ever go onto the lane's slot 0
        // (from an opaque copy of t: hoisted out of the round loop, the flag's thread-constant part is one more register alive
        // across the match phase -- a spill in this kernel)
        uint32_t t_now = t;
        asm volatile("" : "+v"(t_now));
        const uint32_t flag = (t_now & (kSubBytes / 8 - 1)) == 0 ? (kItemRegion | ((rc * kRSubs + (8 * t_now) / kSubBytes) << kItemRegionShift)) : 0u;
        // an item of the chunk: uniform base + a 32-bit byte offset (no 64-bit address arithmetic)
        auto put_item = [&](uint32_t byte_off, uint32_t v) {
          *reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(gi) + (uint64_t)byte_off) = (uint16_t)v;
        };
        // @phase emit.literals trips=1 note=eight slots
        // the literals, slot by slot (a match head only moves the item offset on)
        const uint32_t lits = marks & ~cm;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          if ((lits >> k) & 1) {
            const uint32_t b = ((k < 4 ? B.x : B.y) >> (8 * (k & 3))) & 0xFFu;
            put_item(item_at((1u << k) - 1u), k == 0 ? (kItemTok | b | flag) : (kItemTok | b));
            atomicAdd(&s_hist[b], 1u);
          }
        }
        // @phase emit.matches trips=1 note=two rounds
        // the matches: a lane's eight positions hold at most two taken ones (kMinMatch = 4), so two rounds over
        // the match bits cost less than a match path in each of the eight slots.  Lengths are counted raw (k_plan
        // folds them into symbols)
        static_assert(kMinMatch >= 4, "at most two matches start in eight positions");
        uint32_t rest = cm;
#pragma unroll
        for (uint32_t it = 0; it < 2; ++it) {
          if (rest) {
            const uint32_t k = (uint32_t)__builtin_ctz(rest);
            rest &= rest - 1u;
            const uint32_t at = item_at((1u << k) - 1u);
            uint32_t l3 = (N >> (4 * k)) & 15u;      // capped len-3 from the match phase
            uint32_t d1 = dm1_of(k);
            if (l3 == kCap - 3) {                    // capped match: the walk extended it
              l3 = cap_len - 3;
              d1 = cap_run ? 0u : d1;
            }
            put_item(at, (kItemTok | kItemHead | l3) | (k == 0 ? flag : 0u));
            put_item(at + 2, d1);
            atomicAdd(&s_hist[kHistLen + l3], 1u);
            atomicAdd(&s_hist[kHistD + smem[LD + (d1 < 256 ? d1 : 256 + (d1 >> 7))]], 1u);
          }
        }
      }
      // @phase emit.end trips=1
      tot_tok += rtotal & 0xFFFFu;
      tot_items += (rtotal & 0xFFFFu) + (rtotal >> 16);
      if constexpr (STAMPS) __syncthreads();
      stamp(5);
    }

    // @phase chunk.end trips=0.25 note=once per chunk of four rounds
    // ---- end of a chunk: its counts and histogram go out, the histogram starts over ----
    if (chunk_done || rc == kRoundsPerChunk - 1 || r + 1 == nrounds) {
      __syncthreads();  // this round's histogram updates are complete
      // (a chunk stored by its probe -- taken in one go, no tokens -- has counted nothing and has no histogram: k_plan goes by
      // its token count)
      if (!(chunk_done && tot_tok == 0))
        for (uint32_t idx = t; idx < kHistStride; idx += K1_THREADS) {
          hist_out[(uint64_t)chunk * kHistStride + idx] = s_hist[idx];
          s_hist[idx] = (idx == 256) ? 1u : 0u;
        }
      if (t == 0) { ntok_out[chunk] = tot_tok; nitems_out[chunk] = tot_items | (skip ? kItemsSkipped : 0u); }
      const uint32_t covered = chunk_done ? kSubRegions : (rc + 1) * kRSubs;  // sub-index regions of the rounds that ran
      if (t < kSubRegions && t >= covered) rtok_out[chunk * kSubRegions + t] = tot_tok;
      stamp(6);
    }
    // the next round's first barrier orders this round's LDS reads before its writes
  }
  // @phase (prologue/epilogue) trips=0
  if (nrounds == 0) {  // empty input: one empty chunk
    for (uint32_t idx = t; idx < kHistStride; idx += K1_THREADS) hist_out[(uint64_t)chunk0 * kHistStride + idx] = (idx == 256) ? 1u : 0u;
    if (t == 0) { ntok_out[chunk0] = 0; nitems_out[chunk0] = 0; }
    if (t < kSubRegions) rtok_out[chunk0 * kSubRegions + t] = 0;
  }
  if constexpr (STAMPS) {
    if (t == 0)
      for (int k = 0; k < 8; ++k) stamps[(uint64_t)strip * 8 + k] = st_acc[k];
  }
}

// ---------------------------------------------------------------------------
// K2: per-chunk code plan.  One wave (64-thread workgroup) per chunk.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void or_bits(uint32_t* stage, uint32_t bitpos, uint64_t value) {
  const uint32_t w = bitpos >> 5, sh = bitpos & 31;
  const uint64_t v0 = value << sh;
  const uint32_t v2 = sh ? (uint32_t)(value >> (64 - sh)) : 0u;
  const uint32_t lo = (uint32_t)v0, mid = (uint32_t)(v0 >> 32);
  if (lo) atomicOr(&stage[w], lo);
  if (mid) atomicOr(&stage[w + 1], mid);
  if (v2) atomicOr(&stage[w + 2], v2);
}

// 5 KiB per workgroup (one wave): the CU then holds a chunk in every one of its 32 wave slots, which is what a
// latency-bound kernel wants.  Node weights fit 16 bits (a chunk has at most kChunk tokens + the end-of-block).
// The tree arrays are only fully used by the literal/length tree; what comes after the two big trees (run-length
// items, the code-length code's counts / codes / lengths) lives in their tails -- the code-length tree itself has
// 19 leaves and touches w[0..36], parent[0..36] and key[0..18].
struct PlanSmem {
  uint32_t freq[320];  // ll [0..285], d [288..317]
  uint32_t key[288];   // sorted (freq << 9 | symbol); entries 32.. double as the header bit image (see kHeaderAt)
  union {
    uint16_t w[576];
    struct { uint16_t w_head[64]; uint8_t rle_sym[320]; uint8_t rle_ext[320]; };
  };
  union {
    uint16_t parent[576];
    struct { uint16_t parent_head[128]; uint32_t clfreq[32]; uint32_t cl_code[32]; uint8_t cl_lens[32]; };
  };
  uint32_t cnt[16];    // leaves per depth 1..15; [0] carries the header's bit count at the end
  uint8_t lens[320];   // ll [0..287], d [288..319]
};
static_assert(sizeof(PlanSmem) == 5120, "k_plan: 32 workgroups per CU");
// The header image is built after the literal/length and distance trees are done; the code-length tree that is
// built in between only touches key[0..18].
constexpr uint32_t kHeaderAt = 32;
static_assert(kHeaderAt + kHeaderWords <= 288 && kHeaderAt >= 19, "header image inside key[]");
static_assert(kChunk + 1 <= 65535, "node weights are 16 bits");

// Length-limited Huffman code lengths; all 64 lanes call it.  Specification (DESIGN.md,
// "code lengths"): two-queue Huffman, clamp, Kraft repair, lengths dealt longest-first
// to the rarest symbols.  Three parts, so that the serial one can run elsewhere (k_plan_merge):
//   sort_symbols   -> m used symbols, their keys (freq << 9 | symbol) ascending in S.key[0..m)
//   merge_two_queue: the tree over weights w[0..m) -> parent[0..2m-2), serial in the merges
//   finish_lengths : leaf depths from the parents, clamp + Kraft repair, lengths dealt by rank
template <uint32_t NG>  // 64-symbol groups the alphabet spans: 5 for literal/length, 1 for distance and code-length codes
__device__ uint32_t sort_symbols(PlanSmem& S, const uint32_t* freq, uint32_t n, uint8_t* lens, uint32_t lane) {
  // keys (freq << 9 | symbol) stay in registers: symbol g*64+lane in key[g]
  uint32_t key[NG], rank[NG];
  uint32_t mloc = 0;
#pragma unroll
  for (uint32_t g = 0; g < NG; ++g) {
    const uint32_t s = g * 64 + lane;
    const uint32_t f = s < n ? freq[s] : 0u;
    key[g] = f ? ((f << 9) | s) : 0xFFFFFFFFu;
    rank[g] = 0;
    if (s < n) lens[s] = 0;
    mloc += f != 0;
  }
  if (lane < 16) S.cnt[lane] = 0;
  const uint32_t m = wave_sum(mloc);
  __syncthreads();
  if (m == 0) return 0;
  if (m == 1) {
#pragma unroll
    for (uint32_t g = 0; g < NG; ++g)
      if (key[g] != 0xFFFFFFFFu) lens[g * 64 + lane] = 1;
    __syncthreads();
    return 1;
  }
  // rank sort ascending by (freq, symbol): every used key is broadcast once (v_readlane)
  // and counted by the lanes holding larger keys -- no LDS round trips
#pragma unroll
  for (uint32_t g2 = 0; g2 < NG; ++g2) {
    if (g2 * 64 >= n) break;
    uint64_t used = __ballot(key[g2] != 0xFFFFFFFFu);
    while (used) {
      const uint32_t src = (uint32_t)__builtin_ctzll(used);
      used &= used - 1;
      const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key[g2], (int)src);
#pragma unroll
      for (uint32_t g = 0; g < NG; ++g) rank[g] += kk < key[g];
    }
  }
#pragma unroll
  for (uint32_t g = 0; g < NG; ++g)
    if (key[g] != 0xFFFFFFFFu) S.key[rank[g]] = key[g];
  __syncthreads();
  return m;
}

// two-queue merge (serial; leaves first on ties) by ONE thread: W(k) reads a node's weight, SETW(k, w) writes one,
// PARENT(x, k) records x's parent.  Both queues are consumed in index order, so their next heads are fetched two
// picks ahead of use.  m >= 2.
template <class RdW, class WrW, class WrP>
__device__ __forceinline__ void merge_two_queue(uint32_t m, RdW W, WrW SETW, WrP PARENT) {
  const uint32_t INF = 0xFFFFFFFFu;
  uint32_t i = 0, j = m, k = m;
  uint32_t a0 = W(0), a1 = W(1), a2 = 2 < m ? W(2) : INF;  // leaves i, i+1, i+2
  uint32_t b0 = INF, b1 = INF;                              // internal nodes j, j+1
  while (k < 2 * m - 1) {
    uint32_t x, y, wx, wy;
    if (a0 <= b0) { x = i++; wx = a0; a0 = a1; a1 = a2; a2 = i + 2 < m ? W(i + 2) : INF; }
    else          { x = j++; wx = b0; b0 = b1; b1 = j + 1 < k ? W(j + 1) : INF; }
    if (a0 <= b0) { y = i++; wy = a0; a0 = a1; a1 = a2; a2 = i + 2 < m ? W(i + 2) : INF; }
    else          { y = j++; wy = b0; b0 = b1; b1 = j + 1 < k ? W(j + 1) : INF; }
    const uint32_t sum = wx + wy;
    SETW(k, sum);
    PARENT(x, k);
    PARENT(y, k);
    if (j == k) b0 = sum;          // the new node is the internal queue's head ...
    else if (j + 1 == k) b1 = sum; // ... or its second entry
    ++k;
  }
}

// The same merge for a LANE of its own among 64 (k_plan_merge): no branch on which queue gives the next node -- under
// divergence a wave executes both sides of every such branch anyway, with the exec-mask bookkeeping on top -- but both
// refills asked for and the five queue registers moved by selects.  Same picks, same ties (leaves first), same parents.
template <class RdW, class WrW, class WrP>
__device__ __forceinline__ void merge_two_queue_lane(uint32_t m, RdW W, WrW SETW, WrP PARENT) {
  const uint32_t INF = 0xFFFFFFFFu;
  uint32_t i = 0, j = m, k = m;
  uint32_t a0 = W(0), a1 = W(1), a2 = 2 < m ? W(2) : INF;
  uint32_t b0 = INF, b1 = INF;
  auto pick = [&](uint32_t& x, uint32_t& wx) {
    const bool leaf = a0 <= b0;
    const uint32_t nli = i + 3, nii = j + 2;             // what the queue that is taken from reads next
    const uint32_t nl = W(nli < m ? nli : 0u), ni = W(nii < k ? nii : 0u);
    x = leaf ? i : j;
    wx = leaf ? a0 : b0;
    a0 = leaf ? a1 : a0;
    a1 = leaf ? a2 : a1;
    a2 = leaf ? (nli < m ? nl : INF) : a2;
    b0 = leaf ? b0 : b1;
    b1 = leaf ? b1 : (nii < k ? ni : INF);
    i += leaf ? 1u : 0u;
    j += leaf ? 0u : 1u;
  };
  while (k < 2 * m - 1) {
    uint32_t x, y, wx, wy;
    pick(x, wx);
    pick(y, wy);
    const uint32_t sum = wx + wy;
    SETW(k, sum);
    PARENT(x, k);
    PARENT(y, k);
    b0 = j == k ? sum : b0;                       // the new node is the internal queue's head ...
    b1 = (j != k && j + 1 == k) ? sum : b1;      // ... or its second entry
    ++k;
  }
}

// leaf depths from S.parent (root = node 2m-2), clamped histogram, Kraft repair, lengths dealt longest-first to the
// rarest symbols (S.key[0..m) ascending).  m >= 2; S.cnt zeroed by sort_symbols.
__device__ void finish_lengths(PlanSmem& S, uint32_t m, uint32_t maxbits, uint8_t* lens, uint32_t lane) {
  const uint32_t root = 2 * m - 2;
  for (uint32_t k = lane; k < m; k += 64) {
    uint32_t d = 0, v = k;
    while (v != root) { v = S.parent[v]; ++d; }
    atomicAdd(&S.cnt[d < maxbits ? d : maxbits], 1u);
  }
  __syncthreads();
  if (lane == 0) {
    int32_t over = -(1 << maxbits);
    for (uint32_t l = 1; l <= maxbits; ++l) over += (int32_t)(S.cnt[l] << (maxbits - l));
    while (over > 0) {
      uint32_t l = maxbits - 1;
      while (S.cnt[l] == 0) --l;
      S.cnt[l]--;
      S.cnt[l + 1]++;
      over -= 1 << (maxbits - l - 1);
    }
    while (over < 0) {
      S.cnt[maxbits]--;
      S.cnt[maxbits - 1]++;
      ++over;
    }
  }
  __syncthreads();
  {
    // lengths dealt longest-first to the rarest symbols: cumulative counts once, in registers
    uint32_t cum[16];
    uint32_t c = 0;
#pragma unroll
    for (uint32_t l = 15; l >= 1; --l) {
      c += l <= maxbits ? S.cnt[l] : 0u;
      cum[l] = c;
    }
    for (uint32_t k = lane; k < m; k += 64) {
      uint32_t l = 0;
#pragma unroll
      for (uint32_t q = 1; q <= 15; ++q)
        if (k < cum[q]) l = q;  // cum is non-increasing in q: the last q with k < cum[q] is the largest such length
      lens[S.key[k] & 511u] = (uint8_t)l;
    }
  }
  __syncthreads();
}

template <uint32_t NG>
__device__ void build_lengths(PlanSmem& S, const uint32_t* freq, uint32_t n, uint32_t maxbits,
                              uint8_t* lens, uint32_t lane) {
  const uint32_t m = sort_symbols<NG>(S, freq, n, lens, lane);
  if (m < 2) return;
  for (uint32_t k = lane; k < m; k += 64) S.w[k] = (uint16_t)(S.key[k] >> 9);
  __syncthreads();
  if (lane == 0)
    merge_two_queue(m, [&](uint32_t k) -> uint32_t { return S.w[k]; }, [&](uint32_t k, uint32_t w) { S.w[k] = (uint16_t)w; },
                    [&](uint32_t x, uint32_t k) { S.parent[x] = (uint16_t)k; });
  __syncthreads();
  finish_lengths(S, m, maxbits, lens, lane);
}

// canonical codes (RFC 1951 3.2.2 == huffman::table::canonicalize,
// /root/reference/huffman/src/table.hpp:177-216), stored bit-reversed because the
// decoder shifts code bits in MSB-first (huffman/src/decode.hpp:90-91).
// out[s] = reversed code | len << 16.  n <= 320, all lanes call it.
__device__ void canonical_codes(const uint8_t* lens, uint32_t n, uint32_t* out, uint32_t lane) {
  uint32_t base[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) base[l] = 0;
  uint32_t rank[5], mylen[5];
#pragma unroll
  for (uint32_t g = 0; g < 5; ++g) {
    const uint32_t s = g * 64 + lane;
    const uint32_t len = (g * 64 < n && s < n) ? lens[s] : 0u;
    mylen[g] = len;
    rank[g] = 0;
    if (g * 64 < n) {
#pragma unroll
      for (uint32_t l = 1; l < 16; ++l) {
        const uint64_t mask = __ballot(len == l);
        if (len == l) rank[g] = base[l] + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
        base[l] += (uint32_t)__popcll(mask);
      }
    }
  }
  uint32_t next[16];
  uint32_t code = 0;
  next[0] = 0;
#pragma unroll
  for (int l = 1; l < 16; ++l) {
    code = (code + (l > 1 ? base[l - 1] : 0u)) << 1;
    next[l] = code;
  }
#pragma unroll
  for (uint32_t g = 0; g < 5; ++g) {
    const uint32_t s = g * 64 + lane;
    if (s < n) {
      uint32_t v = 0;
      const uint32_t len = mylen[g];
      if (len) {
        uint32_t nx = 0;
#pragma unroll
        for (uint32_t l = 1; l < 16; ++l)
          if (len == l) nx = next[l];
        const uint32_t c = nx + rank[g];
        v = (__brev(c) >> (32 - len)) | (len << 16);
      }
      out[s] = v;
    }
  }
}

// RLE of one code-length sequence by the whole wave (own state per sequence: the reference
// decoder reads the HLIT and HDIST sequences with separate vectors, src/decompress.cpp:353-360,
// and never bounds-checks a run, :277-296).  Run starts by ballot, run length = distance to the
// next start, items per run in closed form (zeros: 18 x 138, then 18 / 17 / literal zeros;
// other values: the value, 16 x 6, then 16 / literals), item slots by a scan over the starts.
// Writes rle_sym/rle_ext from `base` on, counts symbols into clfreq; returns the item count.
__device__ uint32_t rle_parallel(PlanSmem& S, const uint8_t* lens, uint32_t n, uint32_t base, uint32_t lane) {
  uint64_t Sm[5];
  uint32_t val[5];
#pragma unroll
  for (uint32_t g = 0; g < 5; ++g) {
    const uint32_t i = g * 64 + lane;
    const uint32_t cur = i < n ? lens[i] : 0xFFu;
    const uint32_t prev = (i > 0 && i < n) ? lens[i - 1] : 0xFEu;
    val[g] = cur;
    Sm[g] = __ballot(i < n && (i == 0 || prev != cur));
  }
  uint32_t total = 0;
#pragma unroll
  for (uint32_t g = 0; g < 5; ++g) {
    if (g * 64 >= n) break;
    const uint32_t i = g * 64 + lane;
    const bool start = (Sm[g] >> lane) & 1;
    // next run start after i: in this group, else the first start of a later group, else n
    uint32_t later = n;
#pragma unroll
    for (uint32_t g2 = 4; g2 > g; --g2)
      if (Sm[g2]) later = g2 * 64 + (uint32_t)__builtin_ctzll(Sm[g2]);
    const uint64_t up = lane < 63 ? (Sm[g] >> (lane + 1)) : 0ull;
    uint32_t nxt = up ? i + 1 + (uint32_t)__builtin_ctzll(up) : later;
    nxt = nxt < n ? nxt : n;
    const uint32_t r = start ? nxt - i : 0u;
    const uint32_t v = val[g];
    uint32_t q = 0, rem = 0, cnt = 0;
    if (start) {
      if (v == 0) {
        q = r / 138u;
        rem = r - 138u * q;
        cnt = q + ((rem >= 3) ? 1u : rem);
      } else {
        q = (r - 1) / 6u;
        rem = (r - 1) - 6u * q;
        cnt = 1 + q + ((rem >= 3) ? 1u : rem);
      }
    }
    const uint32_t incl = wave_incl_scan(cnt, lane);
    uint32_t k = base + total + incl - cnt;
    auto put = [&](uint32_t sym, uint32_t ext) {
      S.rle_sym[k] = (uint8_t)sym;
      S.rle_ext[k] = (uint8_t)ext;
      atomicAdd(&S.clfreq[sym], 1u);
      ++k;
    };
    if (start) {
      if (v == 0) {
        for (uint32_t c = 0; c < q; ++c) put(18, 138 - 11);
        if (rem >= 11) put(18, rem - 11);
        else if (rem >= 3) put(17, rem - 3);
        else for (uint32_t c = 0; c < rem; ++c) put(0, 0);
      } else {
        put(v, 0);
        for (uint32_t c = 0; c < q; ++c) put(16, 6 - 3);
        if (rem >= 3) put(16, rem - 3);
        else for (uint32_t c = 0; c < rem; ++c) put(v, 0);
      }
    }
    total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  return total;
}

__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// MODE 0 (k_plan): the whole plan in one launch (rounds 1-5; SFH_PLAN_FUSED=1).  Round 6 -- three launches: MODE 1 (k_plan_sort)
// loads, folds, decides "stored without a code", SORTS both alphabets and leaves keys and weights in PlanTree; k_plan_merge
// builds the trees, one chunk per LANE; MODE 2 (k_plan_finish) picks the parents up, finishes the lengths and does the rest.  The two-queue merge is serial in the
// merges: on lane 0 of a chunk's wave it was 46 % of k_plan's cycles at 1/64 of the lanes -- 58 merges per chunk on text,
// 280 on machine code (1.0 ms per GiB there) -- and with a chunk per lane 64 of them share every instruction.
template <int MODE>
__device__ __forceinline__ void plan_chunk(uint64_t n_total, uint32_t nchunks,
                                           uint32_t* __restrict__ hist, const uint32_t* __restrict__ ntok,
                                           ChunkPlan* __restrict__ plan, ChunkCodes* __restrict__ codes,
                                           uint32_t strategy, uint32_t final_stream,
                                           uint64_t* __restrict__ stamps, PlanTree* __restrict__ ptree) {
  __shared__ PlanSmem S;
  uint32_t* const s_header = S.key + kHeaderAt;
  // diagnostic (stamps != nullptr, SFH_K1_STAMPS=1): cycles per phase at stamps[chunk*8 + 8*nchunks..]
  uint64_t st_t = stamps ? __builtin_amdgcn_s_memtime() : 0;
  uint32_t st_k = 0;
  auto stamp = [&]() {
    if (stamps) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      if (threadIdx.x == 0 && st_k < 8) stamps[((uint64_t)nchunks + blockIdx.x) * 8 + st_k] = now - st_t;
      st_t = now;
      ++st_k;
    }
  };
  const uint32_t lane = threadIdx.x;
  const uint32_t chunk = blockIdx.x;
  const uint64_t cbase = (uint64_t)chunk * kChunk;
  const uint32_t n_raw = (uint32_t)((n_total - cbase) < (uint64_t)kChunk ? (n_total - cbase) : kChunk);
  const bool fin = (chunk + 1 == nchunks) && final_stream;

  [[maybe_unused]] PlanTree* const T = MODE ? ptree + chunk : nullptr;
  if constexpr (MODE == 2) {
    if (T->done) return;  // (uniform: stored without a code by the sorting pass)
  }
  // A chunk with bytes but NO tokens was stored by k_lz77's probe (round 6; probe_span_is_noise in the specification, only ever
  // with strategy 0): it has no histogram -- k_lz77 wrote none -- and nothing to plan
  if (MODE != 2 && strategy == 0 && n_raw != 0 && ntok[chunk] == 0) {  // (uniform)
    ChunkCodes& C0 = codes[chunk];
    for (uint32_t s0 = lane; s0 < 320; s0 += 64) C0.lens[s0] = 0;
    if (lane == 0) {
      C0.header[0] = fin ? 1u : 0u;
      ChunkPlan P;
      P.btype = 0;
      P.out_bytes = n_raw + 5;
      P.header_bits = 3;
      P.body_bits = 0;
      plan[chunk] = P;
      if constexpr (MODE == 1) T->done = 1;
    }
    return;
  }
  // both halves of the chunk's histogram are requested at once (one memory round trip, not two in a row)
  uint32_t rawf[4] = {0, 0, 0, 0};
  if constexpr (MODE != 2) {
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) rawf[q] = hist[(uint64_t)chunk * kHistStride + kHistLen + lane + 64 * q];
  }
  for (uint32_t s = lane; s < 320; s += 64) S.freq[s] = hist[(uint64_t)chunk * kHistStride + s];
  for (uint32_t s = lane; s < 320; s += 64) S.lens[s] = 0;
  __syncthreads();
  if constexpr (MODE != 2) {  // (the finishing pass finds the counts folded: the sorting pass wrote them back)
    // k_lz77 counted match lengths raw (len-3 at kHistLen + 0..255): fold them into the length symbols 257..285
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
      uint32_t eb, ev;
      if (rawf[q]) atomicAdd(&S.freq[len_symbol(lane + 64 * q, eb, ev)], rawf[q]);
    }
    __syncthreads();
    if (lane < 29) hist[(uint64_t)chunk * kHistStride + 257 + lane] = S.freq[257 + lane];  // the folded counts: for parity tests, and for the finishing pass
  }

  stamp();  // 0 load
  if (MODE != 2 && strategy == 0) {
    // A chunk that is (all but) incompressible is STORED without building a code: when the fixed block is no shorter than
    // the stored one and the ESTIMATE of the dynamic block -- the symbols' entropy in fixed point (est_log2: a 64-entry
    // table, the same integers in the specification) + extra bits + the shortest header -- comes within kStoreMargin
    // bytes of it.  Part of the specification (the oracle's plan makes the same choice from the same integers): such a
    // chunk gives up at most kStoreMargin bytes, and the high-entropy chunks of the stored fast path cost a histogram
    // pass instead of three Huffman codes.
    uint32_t fixb = 0, extra = 0, tot = 0;
    for (uint32_t s0 = lane; s0 < 286; s0 += 64) tot += S.freq[s0];
    tot = wave_sum(tot);
    const uint32_t ltot = est_log2(tot);
    uint32_t ent = 0;
    for (uint32_t s0 = lane; s0 < 286; s0 += 64) {
      const uint32_t f = S.freq[s0];
      if (f) ent += f * (ltot - est_log2(f));
      fixb += f * fixed_ll_len(s0);
      if (s0 >= 257) extra += f * len_extra_of_sym(s0 - 257);
    }
    const uint32_t g = lane < 30 ? S.freq[kHistD + lane] : 0u;
    const uint32_t nmat = wave_sum(g);
    if (g) ent += g * (est_log2(nmat) - est_log2(g));
    if (lane < 30) extra += g * dist_extra_of_sym(lane);
    const uint32_t extra_all = wave_sum(extra);
    const uint32_t est_bits = (wave_sum(ent) >> 8) + extra_all + kEstHeaderBits;
    const uint32_t fixbits = 3 + wave_sum(fixb) + extra_all + 5 * nmat;
    const uint32_t est_b = fin ? (est_bits + 7) / 8 : (est_bits + 3 + 7) / 8 + 4;
    const uint32_t fix_b = fin ? (fixbits + 7) / 8 : (fixbits + 3 + 7) / 8 + 4;
    const uint32_t sto_b = n_raw + 5;
    if (fix_b >= sto_b && est_b + kStoreMargin >= sto_b) {
      ChunkCodes& C0 = codes[chunk];
      for (uint32_t s0 = lane; s0 < 320; s0 += 64) C0.lens[s0] = 0;  // (no code was built)
      if (lane == 0) {
        C0.header[0] = fin ? 1u : 0u;
        ChunkPlan P;
        P.btype = 0;
        P.out_bytes = sto_b;
        P.header_bits = 3;
        P.body_bits = 0;
        plan[chunk] = P;
        if constexpr (MODE == 1) T->done = 1;
      }
      return;
    }
  }
  if constexpr (MODE == 0) {
    build_lengths<5>(S, S.freq, 286, 15, S.lens, lane);
    stamp();  // 1 lit/len lengths
    build_lengths<1>(S, S.freq + kHistD, 30, 15, S.lens + 288, lane);
    stamp();  // 2 distance lengths
  } else if constexpr (MODE == 1) {
    // the sorting pass: both alphabets' used symbols ascending by (count, symbol), keys and weights for the merge pass
    const uint32_t m_ll = sort_symbols<5>(S, S.freq, 286, S.lens, lane);
    if (m_ll >= 2)
      for (uint32_t k = lane; k < m_ll; k += 64) {
        const uint32_t key = S.key[k];
        T->key_ll[k] = key;
        T->wt_ll[k] = (uint16_t)(key >> 9);
      }
    __syncthreads();  // (S.key is sorted into again below)
    const uint32_t m_d = sort_symbols<1>(S, S.freq + kHistD, 30, S.lens + 288, lane);
    if (m_d >= 2 && lane < m_d) {
      const uint32_t key = S.key[lane];
      T->key_d[lane] = key;
      T->wt_d[lane] = (uint16_t)(key >> 9);
    }
    if (lane == 0) { T->done = 0; T->m_ll = m_ll; T->m_d = m_d; }
    return;
  } else {
    // the finishing pass: the sorted keys and the merge pass's parents -> lengths (an alphabet of fewer than two used symbols
    // needs no tree: sort_symbols settles it, as in the one-launch kernel)
    const uint32_t m_ll = T->m_ll, m_d = T->m_d;
    if (m_ll < 2) {
      sort_symbols<5>(S, S.freq, 286, S.lens, lane);
    } else {
      if (lane < 16) S.cnt[lane] = 0;
      for (uint32_t k = lane; k < m_ll; k += 64) S.key[k] = T->key_ll[k];
      for (uint32_t v = lane; v < 2 * m_ll - 2; v += 64) S.parent[v] = T->parent_ll[v];
      __syncthreads();
      finish_lengths(S, m_ll, 15, S.lens, lane);
    }
    stamp();  // 1 lit/len lengths
    if (m_d < 2) {
      sort_symbols<1>(S, S.freq + kHistD, 30, S.lens + 288, lane);
    } else {
      if (lane < 16) S.cnt[lane] = 0;
      if (lane < m_d) S.key[lane] = T->key_d[lane];
      if (lane < 2 * m_d - 2) S.parent[lane] = T->parent_d[lane];
      __syncthreads();
      finish_lengths(S, m_d, 15, S.lens + 288, lane);
    }
    stamp();  // 2 distance lengths
  }

  // body costs
  uint32_t dyn = 0, fix = 0, extra = 0, nmatch = 0;
  for (uint32_t s = lane; s < 286; s += 64) {
    const uint32_t f = S.freq[s];
    dyn += f * S.lens[s];
    fix += f * fixed_ll_len(s);
    if (s >= 257) extra += f * len_extra_of_sym(s - 257);
  }
  if (lane < 30) {
    const uint32_t f = S.freq[kHistD + lane];
    dyn += f * S.lens[288 + lane];
    extra += f * dist_extra_of_sym(lane);
    nmatch += f;
  }
  dyn = wave_sum(dyn);
  fix = wave_sum(fix);
  extra = wave_sum(extra);
  nmatch = wave_sum(nmatch);
  const uint32_t dyn_body = dyn + extra;
  const uint32_t fix_body = fix + extra + 5 * nmatch;

  // dynamic header: HLIT/HDIST, RLE, code-length code -- all wave-parallel
  if (lane < 32) S.clfreq[lane] = 0;
  for (uint32_t k = lane; k < kHeaderWords; k += 64) s_header[k] = 0;
  __syncthreads();
  uint32_t hlit, hdist;
  {
    // last used lit/len symbol (EOB = 256 is always used) and distance symbol
    const uint64_t m4 = __ballot(256 + lane < 286 && S.lens[256 + lane] != 0);
    hlit = 256 + 64 - (uint32_t)__builtin_clzll(m4);
    const uint64_t md = __ballot(lane < 30 && S.lens[288 + lane] != 0);
    hdist = md ? 64 - (uint32_t)__builtin_clzll(md) : 1u;
  }
  const uint32_t nl = rle_parallel(S, S.lens, hlit, 0, lane);
  const uint32_t nd = rle_parallel(S, S.lens + 288, hdist, nl, lane);
  const uint32_t nitems = nl + nd;
  __syncthreads();
  stamp();  // 3 costs + RLE
  build_lengths<1>(S, S.clfreq, 19, 7, S.cl_lens, lane);
  canonical_codes(S.cl_lens, 19, S.cl_code, lane);
  __syncthreads();
  stamp();  // 4 code-length code
  {
    const uint64_t mc = __ballot(lane < 19 && S.cl_lens[c_cl_order[lane < 19 ? lane : 0]] != 0);
    uint32_t hclen = mc ? 64 - (uint32_t)__builtin_clzll(mc) : 4u;
    hclen = hclen < 4 ? 4u : hclen;
    // BFINAL, BTYPE=10, HLIT, HDIST, HCLEN: 17 bits; then hclen 3-bit code-length-code lengths
    if (lane == 0)
      or_bits(s_header, 0, (uint64_t)((fin ? 1u : 0u) | (2u << 1) | ((hlit - 257) << 3) | ((hdist - 1) << 8) | ((hclen - 4) << 13)));
    if (lane < hclen) or_bits(s_header, 17 + 3 * lane, (uint64_t)S.cl_lens[c_cl_order[lane]]);
    uint32_t bitbase = 17 + 3 * hclen;
#pragma unroll
    for (uint32_t g = 0; g < 5; ++g) {
      if (g * 64 >= nitems) break;
      const uint32_t k = g * 64 + lane;
      uint32_t nb = 0;
      uint64_t v = 0;
      if (k < nitems) {
        const uint32_t sy = S.rle_sym[k];
        const uint32_t c = S.cl_code[sy];
        const uint32_t cl = c >> 16;
        const uint32_t eb = sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u;
        v = (uint64_t)(c & 0xFFFF) | ((uint64_t)S.rle_ext[k] << cl);
        nb = cl + eb;
      }
      const uint32_t incl = wave_incl_scan(nb, lane);
      if (nb) or_bits(s_header, bitbase + incl - nb, v);
      bitbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) S.cnt[0] = bitbase;  // includes the 3 block-header bits
  }
  __syncthreads();

  stamp();  // 5 header bits
  const uint32_t dyn_hbits = S.cnt[0];
  const uint32_t dyn_bits = dyn_hbits + dyn_body;
  const uint32_t fix_bits = 3 + fix_body;
  const uint32_t dyn_bytes = fin ? (dyn_bits + 7) / 8 : (dyn_bits + 3 + 7) / 8 + 4;
  const uint32_t fix_bytes = fin ? (fix_bits + 7) / 8 : (fix_bits + 3 + 7) / 8 + 4;
  const uint32_t sto_bytes = n_raw + 5;
  uint32_t bt;
  if (strategy == 1) bt = 0;
  else if (strategy == 2) bt = 1;
  else if (strategy == 3) bt = 2;
  else {
    bt = 0;
    uint32_t best = sto_bytes;
    if (fix_bytes < best) { bt = 1; best = fix_bytes; }
    if (dyn_bytes < best) { bt = 2; best = dyn_bytes; }
  }

  ChunkCodes& C = codes[chunk];
  for (uint32_t s = lane; s < 320; s += 64) C.lens[s] = S.lens[s];  // dynamic lengths, for parity tests
  __syncthreads();
  if (bt == 1) {
    for (uint32_t s = lane; s < 320; s += 64) S.lens[s] = s < 288 ? (uint8_t)fixed_ll_len(s) : (uint8_t)5;
    if (lane == 0) s_header[0] = (fin ? 1u : 0u) | (1u << 1);
    __syncthreads();
  } else if (bt == 0) {
    if (lane == 0) s_header[0] = fin ? 1u : 0u;
    __syncthreads();
  }
  canonical_codes(S.lens, 288, C.lcode, lane);
  canonical_codes(S.lens + 288, 32, C.dcode, lane);
  const uint32_t hbits = bt == 2 ? dyn_hbits : 3;
  for (uint32_t k = lane; k < (hbits + 31) / 32; k += 64) C.header[k] = s_header[k];
  if (lane == 0) {
    ChunkPlan P;
    P.btype = bt;
    P.out_bytes = bt == 0 ? sto_bytes : bt == 1 ? fix_bytes : dyn_bytes;
    P.header_bits = hbits;
    P.body_bits = bt == 1 ? fix_body : dyn_body;
    plan[chunk] = P;
  }
  stamp();  // 6 canonical codes + stores
}

#define SF_PLAN_KERNEL(NAME, MODE)                                                                                          \
  __global__ __launch_bounds__(64) void NAME(uint64_t n_total, uint32_t nchunks, uint32_t* __restrict__ hist,                 \
                                             const uint32_t* __restrict__ ntok, ChunkPlan* __restrict__ plan,                 \
                                             ChunkCodes* __restrict__ codes, uint32_t strategy, uint32_t final_stream,        \
                                             uint64_t* __restrict__ stamps, PlanTree* __restrict__ ptree) {                   \
    plan_chunk<MODE>(n_total, nchunks, hist, ntok, plan, codes, strategy, final_stream, stamps, ptree);                       \
  }
SF_PLAN_KERNEL(k_plan, 0)         // one launch (rounds 1-5; SFH_PLAN_FUSED=1)
SF_PLAN_KERNEL(k_plan_sort, 1)    // K2a
SF_PLAN_KERNEL(k_plan_finish, 2)  // K2c
#undef SF_PLAN_KERNEL

// K2b: the two-queue merges of k_plan's three launches, ONE CHUNK PER LANE.  A lane's node weights (u16, leaves then
// internal nodes: 571 at most) live in its own row of LDS -- 289 dwords, an odd stride, so the 64 lanes' accesses at the
// same node index fall on different banks -- and the parents go straight to the chunk's PlanTree (written, never read
// here).  The wave runs as long as its longest merge; on one workload the chunks' alphabets are alike.
constexpr uint32_t KM_ROW = 289;
static_assert(2 * KM_ROW >= 576 && (KM_ROW & 1) == 1 && 64 * KM_ROW * 4 <= 80 * 1024, "k_plan_merge: two waves per CU");
__global__ __launch_bounds__(64) void k_plan_merge(uint32_t nchunks, PlanTree* __restrict__ ptree) {
  __shared__ uint32_t s_rows[64 * KM_ROW];
  const uint32_t lane = threadIdx.x;
  const uint32_t chunk = blockIdx.x * 64 + lane;
  if (chunk >= nchunks) return;  // (no barrier in this kernel: every lane is on its own)
  PlanTree* const T = ptree + chunk;
  if (T->done) return;
  uint16_t* const w = reinterpret_cast<uint16_t*>(s_rows + lane * KM_ROW);
  uint32_t* const w32 = s_rows + lane * KM_ROW;
#pragma unroll 1
  for (uint32_t tree = 0; tree < 2; ++tree) {
    const uint32_t m = tree ? T->m_d : T->m_ll;
    if (m < 2) continue;
    const uint32_t* const wt32 = reinterpret_cast<const uint32_t*>(tree ? T->wt_d : T->wt_ll);  // (4-byte aligned: PlanTree's layout)
    uint16_t* const parent = tree ? T->parent_d : T->parent_ll;
    for (uint32_t k = 0; k < (m + 1) / 2; ++k) w32[k] = wt32[k];  // (the odd last half-word is a weight nobody reads)
    merge_two_queue_lane(m, [&](uint32_t k) -> uint32_t { return w[k]; }, [&](uint32_t k, uint32_t v) { w[k] = (uint16_t)v; },
                         [&](uint32_t x, uint32_t k) { parent[x] = (uint16_t)k; });
  }
}

// ---------------------------------------------------------------------------
// K3: exclusive scan of chunk sizes (single workgroup).
// ---------------------------------------------------------------------------
constexpr uint32_t K3_THREADS = 1024;
constexpr uint32_t K3_TILES = kBatchChunks / K3_THREADS;  // a batch in tiles of one chunk per thread
static_assert(kBatchChunks % K3_THREADS == 0 && K3_TILES == 32 && K3_THREADS / 64 == 16, "k_scan: 32 tiles, a DPP row per tile's wave totals");
static_assert((uint64_t)kBatchChunks * (kChunk + kChunk / 8 + 656) < (1ull << 32), "a batch's bytes fit 32 bits");
// `carry` (a batch after the first): the offsets continue where *total -- the previous batch's end -- left off.
// Round 5: every thread takes chunk t of every tile -- coalesced loads, all 32 in flight at once -- instead of 32 consecutive
// chunks (a 512-byte stride between lanes, one dependent pass for the sums and one for the offsets: 46 us per GiB); the
// tiles' wave totals meet in LDS, one 16-lane DPP row per tile, then the tiles' totals in one wave.
__global__ __launch_bounds__(K3_THREADS) void k_scan(uint32_t nchunks, const ChunkPlan* __restrict__ plan,
                                                     uint64_t base, uint32_t carry, uint64_t* __restrict__ offsets,
                                                     uint64_t* __restrict__ total) {
  __shared__ uint32_t s_wt[K3_TILES][16];  // [tile][wave]: the wave's bytes in the tile, then the bytes of the tile's waves before it
  __shared__ uint32_t s_tile[K3_TILES + 1];  // bytes of the tiles before, [K3_TILES]: of all
  if (carry) base = *total;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t v[K3_TILES], ex[K3_TILES];
#pragma unroll
  for (uint32_t k = 0; k < K3_TILES; ++k) {
    const uint32_t c = k * K3_THREADS + t;
    v[k] = c < nchunks ? plan[c].out_bytes : 0u;
  }
#pragma unroll
  for (uint32_t k = 0; k < K3_TILES; ++k) {
    const uint32_t incl = wave_incl_add(v[k]);
    ex[k] = incl - v[k];
    if (lane == 63) s_wt[k][wave] = incl;
  }
  __syncthreads();
  if (t < K3_TILES * 16) {
    // thread (tile, wave): an inclusive scan inside the tile's row of sixteen (row_shr 1, 2, 4, 8 stay inside a DPP row)
    const uint32_t x = s_wt[t >> 4][t & 15];
    uint32_t r = x;
    r += dpp_from<0x111, 0xF>(r);
    r += dpp_from<0x112, 0xF>(r);
    r += dpp_from<0x114, 0xF>(r);
    r += dpp_from<0x118, 0xF>(r);
    s_wt[t >> 4][t & 15] = r - x;
    if ((t & 15) == 15) s_tile[t >> 4] = r;  // the tile's bytes
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t x = lane < K3_TILES ? s_tile[lane] : 0u;
    const uint32_t r = wave_incl_add(x);
    if (lane < K3_TILES) s_tile[lane] = r - x;
    if (lane == K3_TILES - 1) s_tile[K3_TILES] = r;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < K3_TILES; ++k) {
    const uint32_t c = k * K3_THREADS + t;
    if (c < nchunks) offsets[c] = base + s_tile[k] + s_wt[k][wave] + ex[k];
  }
  if (t == 0) {
    const uint64_t all = base + s_tile[K3_TILES];
    *total = all;
    offsets[nchunks] = all;  // closes the index: chunk c occupies [offsets[c], offsets[c + 1])
  }
}

// ---------------------------------------------------------------------------
// K4: emit.  One workgroup per chunk; bits are OR-ed into an LDS image of the chunk's
// output whose dword k maps onto the aligned global dword (offset>>2)+k.
// ---------------------------------------------------------------------------
constexpr uint32_t K4_THREADS = 512;
constexpr uint32_t K4_WAVES = K4_THREADS / 64;
constexpr uint32_t K4_IPT = 8;  // items per thread per batch (one 16-byte load)
// the image of one chunk's output: sfh_compress_bound's per-chunk share (fixed-Huffman worst case: nine bits per
// literal, + the largest dynamic header) + alignment and the bit writers' slack; four workgroups fit a CU's LDS
constexpr uint32_t K4_STAGE_WORDS = 9392;
static_assert(4 * K4_STAGE_WORDS >= kChunk + kChunk / 8 + 640 + 16 + 8, "k_emit: stage holds the largest chunk");
static_assert(4 * (4 * K4_STAGE_WORDS + 4 * (288 + 32 + 256 + 2 * 8 + kSubRegions) + 512) <= 160 * 1024, "k_emit: four workgroups per CU");

// code bits of one token: literal byte, or match (l3 = len-3, d1 = dist-1)
__device__ __forceinline__ void literal_bits(uint32_t byte, const uint32_t* lcode, uint64_t& value, uint32_t& nb) {
  const uint32_t lc = lcode[byte];
  value = lc & 0xFFFF;
  nb = lc >> 16;
}
// lenlut[l3] = the length code of len-3 with its extra bits appended (value in bits 0..23, bit count in 24..28):
// built once per chunk from the chunk's code, one LDS read per match instead of the symbol arithmetic
__device__ __forceinline__ uint32_t lenlut_entry(uint32_t l3, const uint32_t* lcode) {
  uint32_t le, lv;
  const uint32_t lc = lcode[len_symbol(l3, le, lv)];
  const uint32_t p = lc >> 16;
  return (lc & 0xFFFFu) | (lv << p) | ((p + le) << 24);
}
__device__ __forceinline__ void match_bits(uint32_t l3, uint32_t d1, const uint32_t* lenlut, const uint32_t* dcode,
                                           uint64_t& value, uint32_t& nb) {
  uint32_t de, dv;
  const uint32_t ds = dist_symbol(d1, de, dv);
  const uint32_t le = lenlut[l3], dc = dcode[ds];
  uint32_t p = le >> 24;
  uint64_t v = le & 0xFFFFFFu;
  v |= (uint64_t)(dc & 0xFFFF) << p;
  p += dc >> 16;
  v |= (uint64_t)dv << p;
  p += de;
  value = v;
  nb = p;
}

__global__ __launch_bounds__(K4_THREADS, 4) void k_emit(const uint8_t* __restrict__ src, uint64_t n_total,
                                                     uint32_t /*nchunks*/, uint16_t* items,
                                                     const uint32_t* __restrict__ nitems_in,
                                                     const uint32_t* __restrict__ ntok_in,
                                                     const ChunkPlan* __restrict__ plan,
                                                     const ChunkCodes* __restrict__ codes,
                                                     const uint64_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ rtok,
                                                     uint32_t* __restrict__ subidx,
                                                     uint8_t* __restrict__ dst) {
  // @phase k4.setup trips=0 note=per chunk: tables, header
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[K4_STAGE_WORDS];
  __shared__ uint32_t s_lcode[288];
  __shared__ uint32_t s_dcode[32];
  // ONE table for every kind of item (round 6; before: three tables and four reads per item, every item priced as all
  // three kinds): entry = value (bits 0..19: the code, a length's extra bits appended) | bits of the value << 20 | extra
  // bits to take from the item's own low bits << 25 (a distance: 0..13).  [kTabLit + byte], [kTabLen + len-3], and for a
  // distance - 1 = d: [min(d, 254 + (d >> 7))] -- d itself below 256, one entry per 128 from there on (the symbols from
  // 16 on cover whole multiples of 128)
  __shared__ uint32_t s_tab[1024];
  __shared__ uint32_t s_wtot[2][K4_WAVES];
  __shared__ uint32_t s_rtok[kSubRegions];
  static_assert(K4_THREADS == 512, "one table entry per thread and kind");
  constexpr uint32_t kTabLit = 512, kTabLen = 768, kTabDistN = 510;

  const uint32_t t = threadIdx.x, lane = t & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  const uint32_t chunk = blockIdx.x;
  const ChunkPlan P = plan[chunk];
  const uint64_t off = offsets[chunk];
  const uint64_t cbase = (uint64_t)chunk * kChunk;
  const uint32_t n_raw = (uint32_t)((n_total - cbase) < (uint64_t)kChunk ? (n_total - cbase) : kChunk);
  const ChunkCodes& C = codes[chunk];
  // what a coded chunk needs next, requested before the plan has arrived (a stored chunk does not use it): one memory
  // round trip for the plan and the tables together instead of two in a row
  uint32_t pre_code = t < 288 ? C.lcode[t] : (t < 320 ? C.dcode[t - 288] : 0u);
  uint32_t pre_rtok = t < kSubRegions ? rtok[(uint64_t)chunk * kSubRegions + t] : 0u;
  const uint32_t pre_ntok = ntok_in[chunk], pre_nit = nitems_in[chunk];
  asm volatile("" : "+v"(pre_code), "+v"(pre_rtok));

  uint32_t* const sub = subidx + (uint64_t)chunk * 2 * kSubRegions;  // {bit offset, tokens before} per region
  if (P.btype == 0) {
    // stored block: BFINAL/BTYPE byte, LEN, NLEN, raw bytes (src/decompress.cpp:416-436 inverse)
    if (t < 2 * kSubRegions) sub[t] = 0;
    uint8_t* o = dst + off;
    if (t == 0) {
      o[0] = (uint8_t)(C.header[0] & 1u);
      o[1] = (uint8_t)(n_raw & 0xFF);
      o[2] = (uint8_t)(n_raw >> 8);
      o[3] = (uint8_t)(~n_raw & 0xFF);
      o[4] = (uint8_t)((~n_raw >> 8) & 0xFF);
    }
    uint8_t* d = o + 5;
    const uint8_t* sp = src + cbase;  // 32 KiB aligned
    if (n_raw >= 64) {
      // 16 bytes per thread and step: destination blocks aligned (bytes up to the first one singly), every block from
      // the two aligned source blocks it straddles -- the same byte shift for the whole chunk (uniform), so the dword it
      // starts in is a four-way branch and the rest one v_alignbyte per dword.  The blocks whose second source block
      // would reach past the chunk, and the bytes behind them, go singly at the end (fewer than 48)
      const uint32_t head16 = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
      if (t < head16) d[t] = sp[t];
      const uint32_t sft = head16 & 15u, wsel = sft >> 2, bs = sft & 3u;   // (uniform) source offset of a block mod 16
      const uint32_t nvec = (n_raw - head16) / 16 - (sft ? 1u : 0u);       // blocks whose two source blocks lie inside the chunk
      const uint4* s4 = reinterpret_cast<const uint4*>(sp);
      uint4* d4 = reinterpret_cast<uint4*>(d + head16);
      for (uint32_t k = t; k < nvec; k += K4_THREADS) {
        const uint4 A = s4[k + (head16 >> 4)], B = sft ? s4[k + (head16 >> 4) + 1] : make_uint4(0, 0, 0, 0);
        uint4 r;
        if (wsel == 0) {
          r.x = __builtin_amdgcn_alignbyte(A.y, A.x, bs); r.y = __builtin_amdgcn_alignbyte(A.z, A.y, bs);
          r.z = __builtin_amdgcn_alignbyte(A.w, A.z, bs); r.w = __builtin_amdgcn_alignbyte(B.x, A.w, bs);
        } else if (wsel == 1) {
          r.x = __builtin_amdgcn_alignbyte(A.z, A.y, bs); r.y = __builtin_amdgcn_alignbyte(A.w, A.z, bs);
          r.z = __builtin_amdgcn_alignbyte(B.x, A.w, bs); r.w = __builtin_amdgcn_alignbyte(B.y, B.x, bs);
        } else if (wsel == 2) {
          r.x = __builtin_amdgcn_alignbyte(A.w, A.z, bs); r.y = __builtin_amdgcn_alignbyte(B.x, A.w, bs);
          r.z = __builtin_amdgcn_alignbyte(B.y, B.x, bs); r.w = __builtin_amdgcn_alignbyte(B.z, B.y, bs);
        } else {
          r.x = __builtin_amdgcn_alignbyte(B.x, A.w, bs); r.y = __builtin_amdgcn_alignbyte(B.y, B.x, bs);
          r.z = __builtin_amdgcn_alignbyte(B.z, B.y, bs); r.w = __builtin_amdgcn_alignbyte(B.w, B.z, bs);
        }
        d4[k] = r;
      }
      const uint32_t done16 = head16 + 16 * nvec;
      if (t < n_raw - done16) d[done16 + t] = sp[done16 + t];
      return;
    }
    const uint32_t head = (uint32_t)((4 - ((uintptr_t)d & 3)) & 3);
    const uint32_t h = head < n_raw ? head : n_raw;
    if (t < h) d[t] = sp[t];
    const uint32_t nd = (n_raw - h) / 4;
    uint32_t* d32 = reinterpret_cast<uint32_t*>(d + h);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(sp);
    const uint32_t full_src_words = n_raw / 4;  // dwords entirely inside the chunk
    for (uint32_t k = t; k < nd; k += K4_THREADS) {
      const uint32_t ob = h + 4 * k;  // source byte offset; ob & 3 == h & 3
      const uint32_t w = ob >> 2;
      uint32_t lo = s32[w], hi = 0;
      if ((ob & 3) != 0) {
        if (w + 1 < full_src_words) hi = s32[w + 1];
        else
          for (uint32_t b = 0; b < 4 && 4 * (w + 1) + b < n_raw; ++b) hi |= (uint32_t)sp[4 * (w + 1) + b] << (8 * b);
      }
      d32[k] = __builtin_amdgcn_alignbyte(hi, lo, ob & 3);
    }
    const uint32_t done = h + 4 * nd;
    if (t < n_raw - done) d[done + t] = sp[done + t];
    return;
  }

  const uint32_t sh = (uint32_t)(off & 3);
  const uint32_t nwords = (sh + P.out_bytes + 3) / 4;
  {
    uint4* z = reinterpret_cast<uint4*>(s_stage);
    for (uint32_t k = t; k < (nwords + 2 + 3) / 4 && k < K4_STAGE_WORDS / 4; k += K4_THREADS) z[k] = make_uint4(0, 0, 0, 0);
  }
  if (t < 288) s_lcode[t] = pre_code;  // code | bits << 16
  else if (t < 320) s_dcode[t - 288] = pre_code;
  if (t < kSubRegions) {
    s_rtok[t] = pre_rtok;
    sub[2 * t + 1] = pre_rtok;
  }
  __syncthreads();
  if (t < 256) {
    const uint32_t le = lenlut_entry(t, s_lcode);  // value | bits << 24
    s_tab[kTabLen + t] = (le & 0xFFFFFu) | ((le >> 24) << 20);
    s_tab[kTabLit + t] = (pre_code & 0xFFFFu) | ((pre_code >> 16) << 20);
  }
  if (t < kTabDistN) {
    const uint32_t sym = dist_symbol_of(t < 256 ? t : (t - 254) << 7);
    const uint32_t dc = s_dcode[sym];  // code | length << 16
    s_tab[t] = (dc & 0xFFFFu) | ((dc >> 16) << 20) | (dist_extra_of_sym(sym) << 25);
  } else {
    s_tab[t] = 0;  // (510, 511: where the distance index of a non-distance item may land; never used)
  }
  {
    uint8_t* sb = reinterpret_cast<uint8_t*>(s_stage) + sh;
    const uint8_t* hb = reinterpret_cast<const uint8_t*>(C.header);
    const uint32_t hbytes = (P.header_bits + 7) / 8;
    for (uint32_t k = t; k < hbytes; k += K4_THREADS) sb[k] = hb[k];
  }
  __syncthreads();

  const uint32_t ntok = pre_ntok;
  const uint32_t nit = pre_nit & ~kItemsSkipped;
  // stored fast path: the chunk's items were never written -- they are its bytes (every position a literal), taken from
  // the input here
  const uint16_t* it = items + (uint64_t)chunk * kChunk;
  if (pre_nit & kItemsSkipped) {  // (uniform; rare: a chunk that took the fast path and is NOT stored)
    // write them now, into the chunk's own item slots, and go on as for any chunk (nit == n_raw: every position a literal)
    uint16_t* const wr = items + (uint64_t)chunk * kChunk;
    for (uint32_t pos = t; pos < nit; pos += K4_THREADS) {
      const uint32_t b = src[cbase + pos];
      wr[pos] = (uint16_t)(kItemTok | ((pos & (kSubBytes - 1)) == 0 ? (b | kItemRegion | ((pos / kSubBytes) << kItemRegionShift)) : b));
    }
    __threadfence_block();
    __syncthreads();  // (the stores have completed: every thread reads its items from memory below)
  }
  uint32_t running = 8 * sh + P.header_bits;
  uint32_t buf = 0;
  // items in batches of K4_THREADS*8 (one 16-byte load per thread); the next batch is in flight while this
  // one is packed.  (reads past nit stay inside the chunk's kChunk-slot item area; they are masked below)
  uint4 q_next = make_uint4(0, 0, 0, 0);
  if (nit) q_next = *reinterpret_cast<const uint4*>(it + t * K4_IPT);
  // @phase k4.load trips=1 note=per batch of 4096 items
  for (uint32_t b0 = 0; b0 < nit; b0 += K4_THREADS * K4_IPT, buf ^= 1) {
    const uint32_t i0 = b0 + t * K4_IPT;
    const uint4 q = q_next;
    const bool full = b0 + K4_THREADS * K4_IPT <= nit;  // (uniform) every item of the batch exists
    if (b0 + K4_THREADS * K4_IPT < nit) q_next = *reinterpret_cast<const uint4*>(it + i0 + K4_THREADS * K4_IPT);
    const uint32_t e[K4_IPT] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
    // Every item is ONE unit of at most 28 bits: a literal its code; a match head its length code + extra bits; a
    // distance its code + extra bits.  (A match is not priced as one 48-bit unit: that needs 64-bit shifts and a
    // two-part flush, and the head's thread would have to look at the next thread's item.)  An item says what it is
    // (kItemTok, kItemHead), so its place in the one table is arithmetic on the item alone, and all eight reads of a
    // thread's batch are in flight together.
    uint32_t val[K4_IPT], nb[K4_IPT], mine = 0;
    // @phase k4.lookup trips=1
    uint32_t ent[K4_IPT];
#pragma unroll
    for (uint32_t k = 0; k < K4_IPT; ++k) {
      const uint32_t cur = e[k];
      // a token's first item: byte or len-3 with kItemHead right above it -- its place among the literal and length entries;
      // a distance - 1 (the item itself: bit 15 clear): itself below 256, 254 + (d >> 7) from there on
      static_assert(kItemHead == 0x100 && kTabLen == kTabLit + 0x100 && kTabLit == 0x200, "a token's low nine bits are its table index");
      uint32_t itok = (cur & 0x1FFu) | kTabLit, idist = min(cur, 254u + (cur >> 7));
      asm volatile("" : "+v"(itok), "+v"(idist));  // (both are there: the choice is ONE select -- left alone, the compiler branches around each)
      ent[k] = s_tab[cur >= kItemTok ? itok : idist];
    }
    asm volatile("" : "+v"(ent[0]), "+v"(ent[1]), "+v"(ent[2]), "+v"(ent[3]), "+v"(ent[4]), "+v"(ent[5]), "+v"(ent[6]), "+v"(ent[7]));
    // items that start a token AND carry the region flag (bits 15 and 14), two to a dword: rare (one in ~500)
    auto both = [](uint32_t w) { return w & (w << 1) & 0x80008000u; };
    uint32_t flagged = both(q.x) | both(q.y) | both(q.z) | both(q.w);
    if (!full) {  // (uniform) a chunk's last batch: what lies behind the last item is no item -- an empty entry is no bits
      uint32_t left = nit - i0;  // (wraps to something huge for a thread wholly inside: every k is below it)
      left = i0 < nit ? left : 0u;
#pragma unroll
      for (uint32_t k = 0; k < K4_IPT; ++k) ent[k] = k < left ? ent[k] : 0u;
      flagged = 1;  // ... and the flags are looked at item by item
    }
#pragma unroll
    for (uint32_t k = 0; k < K4_IPT; ++k) {
      // @phase k4.price trips=1
      const uint32_t en = ent[k];
      const uint32_t dl = __builtin_amdgcn_ubfe(en, 20, 5), de = en >> 25;  // bits of the code, extra bits from the item (a distance's)
      val[k] = (__builtin_amdgcn_ubfe(e[k], 0, de) << dl) | (en & 0xFFFFFu);
      nb[k] = dl + de;
      mine += nb[k];
    }
    uint32_t starts = 0;  // bit k: item k starts a token and carries the region flag
    if (flagged) {
      uint32_t i_now = i0;
      asm volatile("" : "+v"(i_now));  // (opaque: the eight compares belong to this rare path, not in front of the loop)
#pragma unroll
      for (uint32_t k = 0; k < K4_IPT; ++k)
        starts |= ((e[k] & (kItemTok | kItemRegion)) == (kItemTok | kItemRegion) && i_now + k < nit) ? 1u << k : 0u;
    }
    // @phase k4.scan trips=1
    const uint32_t incl = wave_incl_scan(mine, lane);
    if (lane == 63) s_wtot[buf][wave] = incl;
    __syncthreads();
    uint32_t pre = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < K4_WAVES; ++w) {
      const uint32_t v = s_wtot[buf][w];
      if (w < wave) pre += v;
      all += v;
    }
    // the thread's codes are contiguous in the stream: gather them in a 64-bit window
    // and OR whole words, instead of one to three atomics per token
    const uint32_t pos = running + pre + incl - mine;
    // @phase k4.subindex trips=1
    if (starts) {
      // k_lz77 flags the first token of every 1024-byte parse region (32 per chunk): its bit offset is the
      // sub-index entry
      uint32_t pk = pos;
#pragma unroll
      for (uint32_t k = 0; k < K4_IPT; ++k) {
        if ((starts >> k) & 1) sub[2 * ((e[k] >> kItemRegionShift) & 31u)] = pk - 8 * sh;
        pk += nb[k];
      }
    }
    // @phase k4.pack trips=1
    uint32_t wi = pos >> 5, ab = pos & 31;
    uint64_t acc = 0;
    auto put = [&](uint32_t v32, uint32_t n) {
      acc |= (uint64_t)v32 << ab;
      ab += n;
      if (ab >= 32) {
        atomicOr(&s_stage[wi++], (uint32_t)acc);
        acc >>= 32;
        ab -= 32;
      }
    };
#pragma unroll
    for (uint32_t k = 0; k < K4_IPT; ++k) put(val[k], nb[k]);
    if (ab) atomicOr(&s_stage[wi], (uint32_t)acc);
    running += all;
  }
  __syncthreads();
  // @phase k4.flush trips=0 note=per chunk
  if (t < kSubRegions && s_rtok[t] >= ntok) sub[2 * t] = running - 8 * sh;  // regions past the data: the end-of-block code
  if (t == 0) {
    // end of block, then (unless this is the stream's final block) an empty stored
    // block 000 / pad / 00 00 FF FF to byte-align the next chunk
    const uint32_t eob = s_lcode[256];
    or_bits(s_stage, running, eob & 0xFFFF);
    const uint32_t endbit = running + (eob >> 16);
    if (!(C.header[0] & 1u)) {
      const uint32_t byte = (endbit + 3 + 7) / 8;  // relative to the stage start (sh included)
      or_bits(s_stage, 8 * (byte + 2), 0xFFFFull);
    }
  }
  __syncthreads();

  // flush: interior dwords coalesced, the two edge dwords bytewise (neighbouring
  // chunks own the other bytes of those dwords)
  uint8_t* gbase = dst + (off - sh);
  uint32_t* g32 = reinterpret_cast<uint32_t*>(gbase);
  const uint32_t lo_b = sh, hi_b = sh + P.out_bytes;
  for (uint32_t w = t; w < nwords; w += K4_THREADS) {
    const uint32_t v = s_stage[w];
    if (4 * w >= lo_b && 4 * w + 4 <= hi_b) {
      g32[w] = v;
    } else {
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t idx = 4 * w + b;
        if (idx >= lo_b && idx < hi_b) gbase[idx] = (uint8_t)(v >> (8 * b));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t init_kernels() { return hipSuccess; }

hipError_t launch_lz77(const uint8_t* src, uint64_t n, uint32_t nchunks, const Workspace& ws,
                       const Options& opt, hipStream_t s) {
  if (opt.strip_bytes == 0 || opt.strip_bytes % kChunk || opt.strip_bytes > kMaxStrip) return hipErrorInvalidValue;
  const uint32_t per = opt.strip_bytes / kChunk;
  const uint32_t nstrips = (nchunks + per - 1) / per;
  const auto launch = [&](auto kernel, uint64_t* stamps) {
    hipLaunchKernelGGL(kernel, dim3(nstrips), dim3(K1_THREADS), 0, s, src, n, opt.strip_bytes, ws.items, ws.nitems, ws.ntok,
                       ws.hist, ws.rtok, opt.lazy, opt.fast_skip, stamps, opt.chain_depth);
  };
  if (opt.recent) {  // exact recency (SFH_EFFORT_RECENT: even positions searched; SFH_EFFORT_RECENT_ALL: every position)
    if (opt.stride2) {
      if (ws.stamps) launch(k_lz77<true, true, true, true, false, false, true>, ws.stamps);
      else launch(k_lz77<false, true, true, true, false, false, true>, (uint64_t*)nullptr);
    } else {
      if (ws.stamps) launch(k_lz77<true, true, true, false, false, false, true>, ws.stamps);
      else launch(k_lz77<false, true, true, false, false, false, true>, (uint64_t*)nullptr);
    }
    return hipGetLastError();
  }
  if (opt.chain_depth) {  // exact hash chains (SFH_EFFORT_BEST / _ULTRA / _EXTREME)
    if (ws.stamps) launch(k_lz77<true, true, true, false, false, true>, ws.stamps);
    else launch(k_lz77<false, true, true, false, false, true>, (uint64_t*)nullptr);
    return hipGetLastError();
  }
  // effort: {even positions: both levels + near, newer level + near, newer level only} {every position: one table, two tables}
  const uint32_t kind = !opt.stride2 ? (opt.long_table ? 4u : 3u) : opt.depth2 ? 0u : (opt.near ? 1u : 2u);
  if (ws.stamps) {
    if (kind == 0) launch(k_lz77<true, true, true, true, false>, ws.stamps);
    else if (kind == 1) launch(k_lz77<true, false, true, true, false>, ws.stamps);
    else if (kind == 2) launch(k_lz77<true, false, false, true, false>, ws.stamps);
    else if (kind == 3) launch(k_lz77<true, true, true, false, false>, ws.stamps);
    else launch(k_lz77<true, true, true, false, true>, ws.stamps);
  } else {
    if (kind == 0) launch(k_lz77<false, true, true, true, false>, (uint64_t*)nullptr);
    else if (kind == 1) launch(k_lz77<false, false, true, true, false>, (uint64_t*)nullptr);
    else if (kind == 2) launch(k_lz77<false, false, false, true, false>, (uint64_t*)nullptr);
    else if (kind == 3) launch(k_lz77<false, true, true, false, false>, (uint64_t*)nullptr);
    else launch(k_lz77<false, true, true, false, true>, (uint64_t*)nullptr);
  }
  return hipGetLastError();
}
hipError_t launch_plan(uint64_t n, uint32_t nchunks, const Workspace& ws, const Options& opt,
                       hipStream_t s) {
  if (opt.plan_fused || !ws.ptree) {
    hipLaunchKernelGGL(k_plan, dim3(nchunks), dim3(64), 0, s, n, nchunks, ws.hist, ws.ntok, ws.plan, ws.codes,
                       opt.strategy, opt.final_stream, ws.stamps, (PlanTree*)nullptr);
    return hipGetLastError();
  }
  // sort (a wave per chunk) -> merge (a lane per chunk) -> finish (a wave per chunk)
  hipLaunchKernelGGL(k_plan_sort, dim3(nchunks), dim3(64), 0, s, n, nchunks, ws.hist, ws.ntok, ws.plan, ws.codes,
                     opt.strategy, opt.final_stream, (uint64_t*)nullptr, ws.ptree);
  hipLaunchKernelGGL(k_plan_merge, dim3((nchunks + 63) / 64), dim3(64), 0, s, nchunks, ws.ptree);
  hipLaunchKernelGGL(k_plan_finish, dim3(nchunks), dim3(64), 0, s, n, nchunks, ws.hist, ws.ntok, ws.plan, ws.codes,
                     opt.strategy, opt.final_stream, ws.stamps, ws.ptree);
  return hipGetLastError();
}
hipError_t launch_scan(uint32_t nchunks, const Workspace& ws, uint64_t base, bool carry, uint64_t* d_total, hipStream_t s) {
  if (nchunks > kBatchChunks) return hipErrorInvalidValue;  // k_scan covers one batch (K3_TILES tiles, 32-bit sums): more would get no offset
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(K3_THREADS), 0, s, nchunks, ws.plan, base, carry ? 1u : 0u, ws.offsets, d_total);
  return hipGetLastError();
}
hipError_t launch_emit(const uint8_t* src, uint64_t n, uint32_t nchunks, const Workspace& ws,
                       uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_emit, dim3(nchunks), dim3(K4_THREADS), 0, s, src, n, nchunks, ws.items, ws.nitems, ws.ntok,
                     ws.plan, ws.codes, ws.offsets, ws.rtok, ws.subidx, dst);
  return hipGetLastError();
}

}  // namespace sf
