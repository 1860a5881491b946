// sf_device.h -- device-side data layout shared by the kernels and the C-ABI host.
//
// Pipeline (one HIP stream, four launches per call):
//   K1 k_lz77   one 1024-thread workgroup per STRIP (block_bytes of input, a whole number of 32 KiB
//               chunks, coded independently of what precedes it): a 32 KiB window + the 8 KiB round in
//               flight + the hash table + per-position (len,dist) of the round in LDS; step-synchronous
//               hash insertion (two history levels per bucket), candidate compare, wave-local
//               register-resident greedy/lazy parse; 16-bit token items + histogram out, one DEFLATE
//               block per chunk.  Other match phases behind the same stage / parse / emit: exact hash chains
//               (SFH_EFFORT_BEST ..: heads + links in LDS, one workgroup per CU) and the step tables filled in
//               position order (SFH_EFFORT_RECENT / _RECENT_ALL: buckets {lo, hi}, the exact predecessor as a
//               candidate); both insert 64 positions per returning LDS atomic and rest on the lane order
//               sf_guard.hip checks at run time
//   K2 k_plan   one wave per chunk: raw length counts folded into symbols, length-limited Huffman lengths (ll, d, cl),
//               canonical codes, dynamic header bits, block type, exact byte size
//   K3 k_scan   exclusive scan of chunk byte sizes -> output offsets, total
//   K4 k_emit   one workgroup per chunk: bit-pack tokens into LDS, flush to the
//               chunk's final byte offset (or copy raw bytes for a stored block)
// With a zlib / gzip container two more launches (sf_checksum.hip):
//   K5 k_checksum  one workgroup per chunk: Adler-32 / CRC-32 partial of the chunk's input bytes
//   K6 k_wrap      one workgroup: fold the partials, write wrapper header + trailer
// A call on more than kBatchChunks chunks runs K1..K4 batch after batch (bounded scratch); the host-buffer entry
// point pipelines smaller batches with their copies.
// The decoder (sf_inflate.hip, sf_inflate_core.h) runs the other way: k_inflate_tokens[_sub] (Huffman codes ->
// tokens, all segments at once), k_inflate_bytes (tokens -> bytes, strip by strip), k_inflate_status.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sf {

constexpr uint32_t kChunk = 32768;      // bytes per DEFLATE block (byte-aligned in the stream)
constexpr uint32_t kWindow = 32768;     // a match reaches back at most this far, and never before its strip
constexpr uint32_t kStep = 1024;        // positions per hash-insertion step (= K1 threads)
constexpr uint32_t kHashBits = 13;      // buckets of two 16-bit history levels each (32 KiB of LDS)
constexpr uint32_t kMaxStrip = 1u << 24;  // largest block_bytes
constexpr uint32_t kRegion = 512;       // parse region: matches never cross it (one wave of k_lz77 parses one)
constexpr uint32_t kSubBytes = 1024;    // sub-index granularity: every kSubBytes-th position starts a token
constexpr uint32_t kCap = 16;           // match-time compare width; longer matches are extended by the parse
constexpr uint32_t kMinMatch = 4;
constexpr uint32_t kFar4 = 4096;        // a match of exactly 4 bytes beyond this distance is not used
constexpr uint32_t kSkipSlack = 128;    // stored fast path: first 8 KiB with >= 8192-128 tokens => no further search
constexpr uint32_t kSkipProbe = 2048;   // ... and behind such a chunk in its strip only this many positions of the 8 KiB are searched
constexpr uint32_t kItemsSkipped = 0x80000000u;  // nitems[chunk]: the items were not written -- they are the chunk's own bytes
                                                 // (stored fast path: every position a literal); k_emit takes them from the input
constexpr uint32_t kSubRegions = kChunk / kSubBytes;  // 32 sub-index entries per chunk
constexpr uint32_t kTokMatch = 0x80000000u;  // decoder token (k_inflate_*): bit31 match, 16..23 len-3, 0..14 dist-1
constexpr uint32_t kTokRegion = 0x40000000u; // decoder token: first token of a parse region, region index in 24..28
// k_lz77 -> k_emit: 16-bit ITEMS, at most kChunk per chunk.  A literal is one item (kItemTok | byte); a match is two:
// head = kItemTok | kItemHead | len-3, then dist-1 (15 bits, bit 15 clear).  Every item says what it is by itself (round 6;
// before, a distance was "the item behind a head" and k_emit looked at every item's neighbour), and kItemHead sits right
// above the byte: a token's low nine bits ARE its place in k_emit's table of literal and length codes.  The first item of a
// parse region's first token also carries kItemRegion and the region's index in bits 9..13 (kItemRegionShift).
constexpr uint32_t kItemTok = 0x8000u;     // a token's first item: a literal or a match head (clear: a distance)
constexpr uint32_t kItemRegion = 0x4000u;
constexpr uint32_t kItemRegionShift = 9;
constexpr uint32_t kItemHead = 0x0100u;    // with kItemTok: a match head (len-3 in bits 0..7)

constexpr uint32_t kChecksumAdler32 = 1;  // = SFH_ZLIB
constexpr uint32_t kChecksumCrc32 = 2;    // = SFH_GZIP
constexpr uint32_t kCrcPoly = 0xEDB88320u;  // RFC 1952 section 8, reflected

constexpr uint32_t kHistStride = 576;   // ll[0..285] at 0, d[0..29] at 288, raw len-3 counts [0..255] at 320
constexpr uint32_t kHistD = 288;
constexpr uint32_t kHistLen = 320;      // k_lz77 counts match lengths raw; k_plan folds them into ll[257..285]
constexpr uint32_t kHeaderWords = 152;  // 608 bytes >= 4495-bit worst-case dynamic header + 3

// per-chunk plan record written by K2, read by K3/K4
struct ChunkPlan {
  uint32_t btype;        // 0 stored, 1 fixed, 2 dynamic
  uint32_t out_bytes;    // exact bytes of this chunk in the stream
  uint32_t header_bits;  // bits in `header` (block header 3 bits + dynamic header)
  uint32_t body_bits;    // token bits + EOB
};
struct ChunkCodes {
  uint32_t lcode[288];   // bit-reversed code | nbits << 16
  uint32_t dcode[32];
  uint32_t header[kHeaderWords];  // LSB-first bitstream: BFINAL,BTYPE, then RFC 1951 3.2.7 header
  uint8_t lens[320];     // ll lens [0..287], d lens [288..319] (debug / parity)
};

// k_plan in three launches (round 6): what the sorting pass hands the merge pass and the merge pass the finishing one, per chunk
struct PlanTree {
  uint32_t done;           // 1: the sorting pass settled the chunk (stored without a code)
  uint32_t m_ll, m_d;      // used literal/length and distance symbols
  uint32_t pad;
  uint32_t key_ll[288];    // (freq << 9 | symbol) ascending, [0 .. m_ll)
  uint32_t key_d[32];
  uint16_t wt_ll[288];     // the same symbols' weights (what the merge reads)
  uint16_t wt_d[32];
  uint16_t parent_ll[576]; // the merge's result: node -> parent, root = node 2m - 2
  uint16_t parent_d[64];
};
static_assert(sizeof(PlanTree) % 16 == 0, "PlanTree rows");

// per-segment record of the decoder (sf_inflate.hip): written by k_inflate_tokens, read by k_inflate_bytes
constexpr uint32_t kSegRaw = 1u, kSegSerial = 2u;
struct SegInfo {
  uint32_t status;   // DecompressStatus of the reference (src/decompress.hpp:13-23), 0 = Success
  uint32_t ntok;
  uint32_t raw;      // bit 0 (kSegRaw): one stored block holding the whole segment, bytes at raw_off of the stream;
                     // bit 1 (kSegSerial): decoded by the lane-serial kernel (diagnostic, SFH_DBG_SEGINFO)
  uint32_t out_n;    // bytes this segment produces
  uint64_t raw_off;
};

// Arrays marked (batch) hold one batch of at most kBatchChunks chunks: a call on a larger input runs the four kernels
// batch after batch on the same stream, so the scratch of a call is bounded (2.2 bytes per input byte of one batch).
constexpr uint32_t kBatchChunks = 32768;  // 1 GiB of input

struct Workspace {
  uint16_t* items;    // (batch) [nchunks][kChunk] token items (K1 -> K4)
  uint32_t* nitems;   // [nchunks]
  uint32_t* tokens;   // [nseg][kChunk] decoder only (k_inflate_tokens* -> k_inflate_bytes)
  uint32_t* ntok;     // [nchunks] tokens (a match counts once)
  uint32_t* hist;     // [nchunks][kHistStride]
  ChunkPlan* plan;    // [nchunks]
  ChunkCodes* codes;  // [nchunks]
  PlanTree* ptree;    // (batch) [nchunks] k_plan's three launches hand their chunk's tree along here
  uint64_t* offsets;  // [nchunks + 1]: first stream byte of every chunk, then the end of the last one (= the index)
  uint64_t* stamps;   // [nchunks][8], diagnostic build only (SFH_K1_STAMPS=1), else null
  uint32_t* sums;     // [nchunks] checksum partials (container modes / sfh_checksum_device)
  SegInfo* seginfo;   // [nchunks] decoder only
  uint32_t* rtok;     // [nchunks][32] tokens before each 1024-byte parse region (K1 -> K4)
  uint32_t* subidx;   // [nchunks][32][2] sub-index: {bit offset of the region's first code, tokens before it}
};

struct Options {
  uint32_t strategy;
  uint32_t final_stream;
  uint32_t lazy;
  uint32_t fast_skip;     // 0: off; 1: stored fast path; 2: ... and a chunk may be stored by its probe (strategy 0: sf_capi.hip)
  uint32_t strip_bytes;  // multiple of kChunk
  uint32_t depth2;       // 1: both history levels of a hash bucket are tried, 0: the newer one only
  uint32_t near;         // 1: the step-local candidate is tried as well (always with depth2)
  uint32_t stride2;      // 1: only even positions are searched, odd ones take over their successor's match (0: thorough)
  uint32_t long_table;   // 1: two tables of 4096 buckets, keyed by four and by seven bytes (with stride2 = 0: SFH_EFFORT_MAX)
  uint32_t chain_depth;  // > 0: exact hash chains of this depth instead of the step tables (SFH_EFFORT_BEST 8, _ULTRA 16, _EXTREME 32)
  uint32_t recent;       // 1: exact recency (SFH_EFFORT_RECENT): buckets {latest, the one before the latest inserting step} + the exact predecessor
  uint32_t plan_fused;   // 1: k_plan as ONE launch, its merges on lane 0 of every chunk's wave (the rounds 1-5 kernel; SFH_PLAN_FUSED=1, for A/B)
};

hipError_t launch_lz77(const uint8_t* src, uint64_t n, uint32_t nchunks, const Workspace& ws,
                       const Options& opt, hipStream_t s);
hipError_t launch_plan(uint64_t n, uint32_t nchunks, const Workspace& ws, const Options& opt,
                       hipStream_t s);
// offsets start at `base` (bytes of wrapper header in front of the stream), or with `carry` at the current *d_total
// (the end of the previous batch); *d_total = the end of this batch
hipError_t launch_scan(uint32_t nchunks, const Workspace& ws, uint64_t base, bool carry, uint64_t* d_total, hipStream_t s);
hipError_t launch_emit(const uint8_t* src, uint64_t n, uint32_t nchunks, const Workspace& ws,
                       uint8_t* dst, hipStream_t s);
hipError_t init_kernels();

// sf_checksum.hip
uint32_t wrapper_header_bytes(uint32_t kind);
hipError_t launch_checksum(const uint8_t* src, uint64_t n, uint32_t nchunks, uint32_t kind, uint32_t* sums,
                           hipStream_t s);
// dst != null: header at dst[0..), trailer at dst[*d_total..), *d_total += trailer bytes; d_value (nullable) = checksum
hipError_t launch_wrap(const uint32_t* sums, uint32_t nchunks, uint64_t n, uint32_t kind, uint8_t* dst,
                       uint64_t* d_total, uint32_t* d_value, hipStream_t s);
// sf_inflate.hip
hipError_t init_inflate_kernels();
// sps: segments per strip (1: every segment independent); a segment's matches may reach its strip's earlier segments
hipError_t launch_inflate_tokens(const uint8_t* src, uint64_t src_n, const uint64_t* index, uint32_t nseg, uint64_t dst_n,
                                 uint32_t* tokens, SegInfo* info, uint32_t sps, bool speculate, hipStream_t s);
hipError_t launch_inflate_tokens_sub(const uint8_t* src, uint64_t src_n, const uint64_t* index, const uint32_t* subidx,
                                     uint32_t nseg, uint64_t dst_n, uint32_t* tokens, SegInfo* info, uint32_t sps,
                                     hipStream_t s);
hipError_t launch_inflate_bytes(const uint8_t* src, uint64_t src_n, uint32_t nseg, const uint32_t* tokens, SegInfo* info,
                                uint8_t* dst, uint32_t sps, hipStream_t s);
hipError_t launch_inflate_status(const SegInfo* info, uint32_t nseg, uint32_t* d_result, hipStream_t s);

// sf_guard.hip: does the LDS execute a returning atomic's lanes in ascending order (op 0: ds_wrxchg_rtn_b32, 1: ds_mskor_rtn_b32)?
// d_result[0] = mismatches against the sequential model, [1] = positions checked
hipError_t run_lds_order_check(int op, uint32_t blocks, uint32_t iters, uint32_t* d_result, hipStream_t s);

uint32_t crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);        // host arithmetic
uint32_t adler32_combine(uint32_t adler_a, uint32_t adler_b, uint64_t len_b);  // host arithmetic

}  // namespace sf
