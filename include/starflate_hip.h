/*
 * starflate_hip.h -- C-ABI of the MI355X DEFLATE compressor (libstarflate_hip.so).
 *
 * The reference (garymm/starflate) has no FFI/plugin interface and no
 * compressor (README.md:5-7).  The boundary kept here is the shape of its one
 * public function,
 *     starflate::decompress(span<const byte> src, span<byte> dst) -> DecompressStatus
 *     (/root/reference/src/decompress.hpp:63-71):
 * caller-owned buffers in, callee never allocates output, no exceptions, a small
 * integer status out.  By default sfh_compress* produce raw RFC 1951 streams (no
 * zlib/gzip wrapper, as /root/reference/tools/deflate_compress.py:8-13 does for the
 * reference's fixtures) that the reference's decompress() inverts; the wrappers
 * that tool strips (RFC 1950 / RFC 1952) are available through sfh_options.container.
 * The C++23 wrapper starflate::compress() (include/starflate/compress.hpp) is the
 * only intended caller besides tests and bench.py (ctypes).
 *
 * Threading: a ctx is not thread-safe; distinct ctxs are independent.  A ctx owns
 * its device scratch; all other buffers are caller-owned.
 */
#ifndef STARFLATE_HIP_H
#define STARFLATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfh_ctx sfh_ctx;

/* negative return values of every sfh_* function returning int */
enum sfh_status {
  SFH_OK = 0,
  SFH_E_INVALID_ARG = -1,   /* null pointer, bad option value, misaligned device pointer */
  SFH_E_DST_TOO_SMALL = -2, /* cap < sfh_compress_bound(n) -- mirrors DecompressStatus::DstTooSmall */
  SFH_E_NO_DEVICE = -3,     /* no HIP device / device index out of range */
  SFH_E_HIP = -4,           /* a HIP runtime call failed; see sfh_last_error() */
  SFH_E_NOMEM = -5,         /* device scratch allocation failed */
  SFH_E_COMM = -6,          /* RCCL not loadable, or an RCCL call failed; see sfh_last_error() */
  SFH_E_UNSUPPORTED = -7    /* the effort asked for rests on LDS behaviour this device does not show (sfh_lds_order_check) */
};

/* block strategy (inverse of src/decompress.cpp:416-458 dispatch) */
enum sfh_strategy {
  SFH_AUTO = 0,   /* per chunk: smallest of stored / fixed / dynamic */
  SFH_STORED = 1, /* BTYPE 00 only  (src/decompress.cpp:416-436) */
  SFH_FIXED = 2,  /* BTYPE 01 only  (src/decompress.cpp:437-446) */
  SFH_DYNAMIC = 3 /* BTYPE 10 only  (src/decompress.cpp:447-458) */
};

/* stream wrapper around the raw DEFLATE data (what tools/deflate_compress.py:8-13 removes) */
enum sfh_container {
  SFH_RAW = 0,  /* RFC 1951 only: what the reference's decompress() reads */
  SFH_ZLIB = 1, /* RFC 1950: 78 9C, stream, Adler-32 big-endian */
  SFH_GZIP = 2  /* RFC 1952: 1F 8B 08 00, MTIME 0, XFL 0, OS 255, stream, CRC-32, ISIZE (both little-endian) */
};

typedef struct sfh_options {
  uint32_t strategy;     /* enum sfh_strategy */
  uint32_t final_stream; /* 1: last block carries BFINAL (src/decompress.cpp:410-415);
                            0: stream ends byte-aligned and non-final (a GPU shard
                            that is not the last one) */
  uint32_t lazy;         /* 0..3: lazy match deferral, positions of look-ahead (default 3) */
  uint32_t no_stored_fast_path; /* 0 (default): a 32 KiB chunk whose first 8 KiB parse to (almost) only
                            literals is not searched further and is coded as literals throughout (high-entropy
                            data -> stored blocks); the chunk behind it in its strip is probed on its first
                            2 KiB only (a sixteenth of the match work while the data stays like that); with
                            SFH_STRATEGY_AUTO a full chunk whose first 8 KiB of BYTES are as good as uniform (their
                            entropy within 64 bytes of 8 KiB) is stored outright, its other 24 KiB never fetched by the
                            match kernel (round 6);
                            1: always search the whole chunk */
  uint32_t container;    /* enum sfh_container; SFH_ZLIB / SFH_GZIP need final_stream = 1.  The checksum is
                            computed on the GPU from the same device buffer (two more launches) */
  uint32_t block_bytes;  /* bytes of input coded independently of what precedes them (a "strip"): a multiple of
                            32768 up to 16 MiB; 0 (default) = SFH_DEFAULT_BLOCK_BYTES, more (SFH_LARGE_BLOCK_BYTES,
                            SFH_CHAIN_BLOCK_BYTES with the chain efforts) for inputs of half a GiB and up, less for inputs
                            too small to fill the device with strips of that size (a function of n and the effort alone).  A strip is written
                            as one byte-aligned DEFLATE block per 32 KiB; inside it the 32 KiB window slides across
                            those blocks (src/decompress.cpp:178 only requires distance <= bytes written), so
                            larger strips compress better; 32768 makes every DEFLATE block independent */
  uint32_t effort;       /* enum sfh_effort.  SFH_EFFORT_DEFAULT searches the even positions (an odd one takes over its
                            successor's match when its own byte fits in front of it) with both history levels of a hash
                            bucket plus the step-local candidate; SFH_EFFORT_FAST only the newer level (about 2 % more
                            output); SFH_EFFORT_FASTEST drops the step-local candidate as well (about 4 % more than the
                            default on text, more on very repetitive data); SFH_EFFORT_THOROUGH searches every
                            position, in insertion steps of 512 (about 1.3 % less output than the default on text, 2.5 %
                            on mixed data, for a quarter more time); SFH_EFFORT_MAX adds a second hash table keyed by seven
                            bytes to that (two tables of 4096 buckets instead of one of 8192: four far candidates per
                            position; 2.7 % less output than thorough on text, for a third more time);
                            SFH_EFFORT_BEST / _ULTRA / _EXTREME replace the step tables by exact HASH CHAINS, zlib's own
                            structure (every position inserted and searched, the 8 / 16 / 32 most recent positions with
                            its hash tried, nearest first): what closes the gap to zlib -6 on real data, where recency
                            counts for more than on the synthetic text, at a quarter to a tenth of the default's speed;
                            EXTREME is zlib -6's own ratio on the text workload.
                            SFH_EFFORT_RECENT / SFH_EFFORT_RECENT_ALL keep the step tables' search pattern (every other
                            position / every position, three candidates each) but fill the buckets in POSITION ORDER:
                            a bucket holds the latest position with the hash and the one before the latest inserting
                            step, and a position's third candidate is its exact predecessor -- the nearest earlier
                            position with its hash, inside the step or before it.  RECENT_ALL is thorough's pattern (on
                            real bytes max's ratio or better -- machine code +3 points -- for a tenth less time): the
                            effort for real source text and machine code.  RECENT (the default's pattern) is DEPRECATED:
                            measured, it is thorough's ratio at thorough's speed on every workload (round 5), i.e. no
                            point of its own on the speed / ratio curve; it stays accepted and bit-exact, new callers use
                            THOROUGH or RECENT_ALL.  Both rest on the LDS executing the lanes of one returning atomic in
                            ascending order (sfh_lds_order_check) */
  uint32_t chain_depth;  /* 0: what the effort implies.  With a chain effort (SFH_EFFORT_BEST / _ULTRA / _EXTREME) any depth
                            1..255 -- candidates per position, most recent first (the specification's chain_depth): 4 is
                            SFH_EFFORT_MAX's ratio on text and the chains' on real data at 80 K MiB/s.  Must be 0 with the
                            table efforts (this was the `reserved` word: a caller that zeroes it gets what it got) */
} sfh_options;

enum sfh_effort { SFH_EFFORT_DEFAULT = 0, SFH_EFFORT_FAST = 1, SFH_EFFORT_FASTEST = 2, SFH_EFFORT_THOROUGH = 3, SFH_EFFORT_MAX = 4,
                  SFH_EFFORT_BEST = 5, SFH_EFFORT_ULTRA = 6, SFH_EFFORT_EXTREME = 7, SFH_EFFORT_RECENT = 8, SFH_EFFORT_RECENT_ALL = 9 };
/* (SFH_EFFORT_RECENT is deprecated: see the `effort` field above) */

#define SFH_DEFAULT_BLOCK_BYTES 262144u
/* block_bytes = 0 on inputs large enough to fill the device four times over with strips of that size (2048 / 1024 of them):
 * 512 KiB, 1 MiB with SFH_EFFORT_BEST and above */
#define SFH_LARGE_BLOCK_BYTES 524288u
#define SFH_CHAIN_BLOCK_BYTES 1048576u

/* fills *o with defaults: AUTO, final_stream=1, lazy=3, block_bytes=0, effort=SFH_EFFORT_DEFAULT */
void sfh_default_options(sfh_options* o);

int sfh_device_count(void);

/* what hipDeviceProp_t reports for `device` (bench.py prints the HBM peak these imply) */
typedef struct sfh_device_props {
  char name[64];
  char arch[32];             /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
  uint32_t compute_units;
  uint32_t lds_bytes_per_cu; /* maxSharedMemoryPerMultiProcessor */
  uint32_t l2_bytes;
  uint32_t memory_clock_khz;
  uint32_t memory_bus_bits;
  uint32_t clock_khz;        /* clockRate: the shader clock bench.py turns kernel times into cycles with */
  uint64_t total_memory;
} sfh_device_props;
int sfh_get_device_props(int device, sfh_device_props* out);
/* Environment read by sfh_create (diagnostics and tests; none changes a stream): SFH_BATCH_CHUNKS=<n> (32 KiB chunks per batch of
 * the compressor and the decoder, default 32768 = 1 GiB), SFH_PLAN_FUSED=1 (the code-length stage as ONE launch, k_plan, instead
 * of k_plan_sort / k_plan_merge / k_plan_finish), SFH_INFLATE_SERIAL=1, SFH_K1_STAMPS=1, SFH_FORCE_ORDER_FAIL=1 (below). */
int sfh_create(sfh_ctx** out, int device);
/* The chain efforts (SFH_EFFORT_BEST / _ULTRA / _EXTREME) and SFH_EFFORT_RECENT insert 64 positions into their hash buckets
 * with ONE returning LDS atomic and rely on the LDS executing that wave-instruction's lanes in ascending order where they
 * meet at one address (op 0: ds_wrxchg_rtn_b32, op 1: ds_mskor_rtn_b32) -- measured behaviour of gfx950, not an ISA
 * promise.  This runs the check the library itself runs ONCE PER CONTEXT, inside sfh_create (two launches of well under a
 * millisecond on the context's own stream and one synchronisation there; the verdict is cached, so no compress call -- the
 * asynchronous ones included -- ever blocks or launches anything for it):
 * `blocks` (1..4096) workgroups x `iters` (1..128) insertion steps x five collision densities (the kernel counts in 32 bits:
 * larger arguments are SFH_E_INVALID_ARG) in the match kernel's own access pattern
 * (partial exec masks, sixteen back-to-back instructions by one wave on shared buckets, the other waves reading the
 * table meanwhile), every position compared with the sequential model.  *mismatches == 0: the order holds.  A compress
 * call with one of those efforts on a context whose check failed returns SFH_E_UNSUPPORTED and changes nothing
 * (sfh_last_error says why); every other effort is unaffected.  SFH_FORCE_ORDER_FAIL=1 in the environment at
 * sfh_create makes the library's own check report failure (tests of that branch). */
int sfh_lds_order_check(sfh_ctx* ctx, uint32_t op, uint32_t blocks, uint32_t iters, uint64_t* mismatches, uint64_t* checked);
void sfh_destroy(sfh_ctx* ctx);
const char* sfh_last_error(const sfh_ctx* ctx);

/* worst-case output bytes for n input bytes (any strategy, any container); block_bytes as in sfh_options
 * (SURVEY.md 8(b) signature).  The bound is per 32 KiB DEFLATE block and a strip is a whole number of those, so every
 * valid block_bytes gives the same figure; a block_bytes the compress calls reject (not a multiple of 32768, or above
 * 16 MiB) returns 0. */
size_t sfh_compress_bound(size_t n, uint32_t block_bytes);

/* Host buffers, synchronous.  *out_n = stream bytes.  Inside the call the input goes up, through the kernels and the
 * stream comes down in 64 MiB batches on three streams, so with pinned buffers the call takes about as long as the
 * larger of its two copies (pageable buffers make the copies themselves synchronous). */
int sfh_compress(sfh_ctx* ctx, const void* src, size_t n, void* dst, size_t cap, size_t* out_n,
                 const sfh_options* opt);

/* One process, several GPUs: `src` is cut into nctx contiguous shards on 32 KiB boundaries,
 * shard i is compressed by ctxs[i] (normally one ctx per device; the contexts must be distinct) from its own host
 * thread, every shard but the last as a non-final byte-aligned stream, and the streams are concatenated in `dst`
 * -- one valid stream, bit-identical to what a single sfh_compress call writes.  cap >= sfh_compress_bound(n).
 * opt->final_stream applies to the last shard; a container wraps the whole (shard checksums are computed on their
 * devices and combined).  The block index of the whole stream is not assembled (use the per-ctx ones). */
int sfh_compress_multi(sfh_ctx* const* ctxs, int nctx, const void* src, size_t n, void* dst, size_t cap,
                       size_t* out_n, const sfh_options* opt);

/* One process PER GPU (north_star: "RCCL over xGMI only to concatenate the independently-encoded
 * block streams"): every rank compresses its own shard -- whole strips, final_stream = 0 on every rank but the last -- and
 * the byte-aligned streams are put back to back on rank `root`.  No data-path collective: shard legality is
 * /root/reference/src/decompress.cpp:178 (a match only has to stay inside the bytes already written) and :410-415
 * (blocks until BFINAL).
 *
 * sfh_gather_offsets: the host arithmetic.  offsets[r] = first byte of rank r's stream in the concatenation that starts
 * at `base`, offsets[nranks] = its end; SFH_E_DST_TOO_SMALL when that exceeds cap, SFH_E_INVALID_ARG on overflow.
 *
 * sfh_gather_streams: called by EVERY rank of `nccl_comm` (an ncclComm_t of RCCL, passed as a plain pointer; RCCL is bound
 * with dlopen at the first call, so the library itself does not link it).  d_stream / d_size: this rank's stream and its
 * byte count ON THE DEVICE, as sfh_compress_device_async leaves them.  One ncclAllGather of the u64 sizes, one read-back of
 * them (h_sizes[nranks], every rank: the transfers need host counts), then ONE grouped round of ncclSend / ncclRecv --
 * each peer's bytes travel once, point to point over their own xGMI link, straight to d_out + offsets[peer] on the
 * root; the root's own stream is a device copy (none when it already lies at its place).  Everything is enqueued on
 * `stream`, which is synchronised once, for the sizes; the call returns with the transfers in flight.  *out_end (every
 * rank) = end of the concatenation.  d_out is only read on the root; `base` and `cap` (the root's: bytes already in d_out,
 * its capacity) are passed alike by every rank, so that all ranks refuse together -- before anything is posted -- when the
 * streams do not fit.  The exchange carries, beside the size, every rank's base and cap and the root's d_stream / d_out
 * addresses (five u64 per rank): ranks that DISAGREE on base or cap, and a root whose d_stream overlaps the gathered range
 * of d_out without lying exactly at its own place there (a peer's bytes would land on it, or it would be copied onto
 * itself), are SFH_E_INVALID_ARG on EVERY rank alike, still before anything is posted.
 * Contract for what the library cannot see: a failure that strikes ONE rank after the exchange -- the read-back or the
 * stream synchronisation failing, ncclSend / ncclRecv returning an error -- returns on that rank only; its peers are then
 * inside a grouped transfer that cannot complete.  The communicator is broken at that point (as after any failed RCCL
 * call): abort it (ncclCommAbort) on all ranks and build a new one. */
int sfh_gather_offsets(const uint64_t* sizes, int nranks, uint64_t base, uint64_t cap, uint64_t* offsets);
/* ranks of the communicator and this process's rank in it (ncclCommCount / ncclCommUserRank through the same binding) */
int sfh_comm_ranks(void* nccl_comm, int* nranks, int* rank);
int sfh_gather_streams(sfh_ctx* ctx, void* nccl_comm, int root, const void* d_stream, const uint64_t* d_size, void* d_out,
                       uint64_t base, uint64_t cap, uint64_t* h_sizes, uint64_t* out_end, void* stream);

/* Device buffers (d_src 16-byte aligned), enqueued on `stream` (a hipStream_t,
 * NULL = the ctx's own stream); synchronises the stream and returns the size. */
int sfh_compress_device(sfh_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t cap,
                        size_t* out_n, const sfh_options* opt, void* stream);

/* Same, but only enqueues: no host synchronisation.  The stream size is left in
 * device memory at *d_out_n (uint64_t, device pointer, may not be NULL). */
int sfh_compress_device_async(sfh_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t cap,
                              uint64_t* d_out_n, const sfh_options* opt, void* stream);

/* ---- measurement hooks (bench.py, tests) ---- */

/* ---- block index + GPU decompress ----
 * A stream written by sfh_compress* is a sequence of byte-aligned segments, one DEFLATE block per 32 KiB of input
 * (SFH_SEGMENT_BYTES).  The index is the table of their first bytes plus the end of the last one: nseg + 1
 * uint64 offsets into the stream (wrapper header included, trailer excluded).  Segments group into strips of
 * block_bytes of input (sfh_options.block_bytes, sfh_last_block_bytes): no match reaches before its strip, so
 * strips decode independently; inside a strip the Huffman decoding of all segments still runs at once and only
 * the byte copies go segment by segment.  With the index the reference's decompress()
 * (src/decompress.hpp:63-71) runs on the GPU; without it DEFLATE decoding is serial (README.md:5-6).  Any
 * stream with such an index qualifies, e.g. zlib output flushed with Z_FULL_FLUSH every 32 KiB (block_bytes =
 * 32768: every segment independent). */
#define SFH_SEGMENT_BYTES 32768u

/* strip size (sfh_options.block_bytes after defaulting) of the last sfh_compress* call on this ctx; 0 before any */
uint32_t sfh_last_block_bytes(const sfh_ctx* ctx);

/* entries of the index of the last sfh_compress* call on this ctx (segments + 1); 0 before any call */
size_t sfh_index_entries(const sfh_ctx* ctx);
/* copies that index to `dst` (host memory, or device memory if dst_on_device); synchronises `stream` */
int sfh_copy_index(sfh_ctx* ctx, uint64_t* dst, size_t entries, int dst_on_device, void* stream);

/* Sub-index of the last sfh_compress* call: per segment, for each of its 32 regions of 1024 bytes of output
 * (no match of this library's streams crosses them) {bit offset of the region's first token code counted from
 * the segment's first byte, tokens before the region}: SFH_SUBINDEX_WORDS uint32 per segment, all zero for a
 * stored segment.  Optional side information: with it the 32 lanes that decode one segment's Huffman codes side
 * by side are TOLD where their regions start (the decoder checks it against the stream: a wrong sub-index is an
 * error, never wrong output).  Streams from elsewhere have none: their segments are decoded by 32 lanes as well,
 * which find their token boundaries speculatively, block after block (up to four blocks with output per segment
 * plus the empty stored blocks a flush leaves; about 1.5 x the sub-indexed time), and by one lane per segment where
 * a segment is cut into more blocks or is damaged -- the serial decoder itself, statuses included
 * (SFH_INFLATE_SERIAL=1 in the environment of sfh_create: that kernel for every segment). */
#define SFH_SUBINDEX_WORDS 64u
int sfh_copy_subindex(sfh_ctx* ctx, uint32_t* dst, size_t words, int dst_on_device, void* stream);

/* Device buffers (d_src 4-byte, d_index 8-byte, d_dst 16-byte aligned; d_subindex 4-byte aligned or NULL;
 * d_src readable up to the next multiple of 4 bytes, as any device allocation is).
 * nseg must be ceil(dst_n / 32768)
 * (1 for dst_n = 0); segment i decodes stream bytes [index[i], index[i+1]) into dst[i*32768 ...) and must
 * produce exactly that many bytes.  block_bytes: the strip size the stream was written with (a multiple of 32768;
 * 0 = 32768, every segment independent): a match of segment i may reach back to the first byte of its strip,
 * a farther one is InvalidDistance.  Returns SFH_OK when the kernels ran; *status is then the reference's
 * DecompressStatus (0 = Success) of the first failing segment in stream order; dst is complete only on 0.
 * Synchronises `stream` (NULL = the ctx's own).
 * Scratch: the decoder keeps 4 bytes of tokens per output byte -- of ONE batch of whole strips, at most 1 GiB of output (round
 * 6; before: of the whole call): 4 GiB at most whatever dst_n is (sfh_last_decode_scratch_bytes), the two kernels alternating
 * batch after batch on the stream; bytes and status are those of one pass over everything
 * (/root/reference/src/decompress.cpp:197-242 semantics: the first failing block in stream order). */
int sfh_decompress_device(sfh_ctx* ctx, const void* d_src, size_t src_n, const uint64_t* d_index,
                          const uint32_t* d_subindex, size_t nseg, void* d_dst, size_t dst_n, uint32_t block_bytes,
                          uint32_t* status, void* stream);
/* Host buffers: H2D (stream + index [+ sub-index, may be NULL]), decode, D2H. */
int sfh_decompress(sfh_ctx* ctx, const void* src, size_t src_n, const uint64_t* index, const uint32_t* subindex,
                   size_t nseg, void* dst, size_t dst_n, uint32_t block_bytes, uint32_t* status);

/* bytes of decoder token scratch the last sfh_decompress* call on this ctx used (0 before the first) */
size_t sfh_last_decode_scratch_bytes(const sfh_ctx* ctx);

#define SFH_INFLATE_NSTAGES 2 /* 0 k_inflate_tokens (Huffman decode), 1 k_inflate_bytes (match copies) */
/* with profiling on: milliseconds per decoder kernel of the last sfh_decompress* call (summed over its batches) */
int sfh_last_inflate_ms(sfh_ctx* ctx, float ms[SFH_INFLATE_NSTAGES]);
const char* sfh_inflate_stage_name(int stage);

/* ---- container checksums ---- */

/* Checksum of n device bytes (d_src 16-byte aligned): kind = SFH_ZLIB -> Adler-32, SFH_GZIP -> CRC-32,
 * bit-exact with zlib's adler32()/crc32().  Synchronises `stream` (NULL = the ctx's own). */
int sfh_checksum_device(sfh_ctx* ctx, const void* d_src, size_t n, uint32_t kind, uint32_t* out, void* stream);
/* Checksum of A||B from the checksums of A and B and the length of B (host arithmetic; a multi-GPU job
 * combines its shards' checksums with these and writes one wrapper around the concatenated raw streams). */
uint32_t sfh_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
uint32_t sfh_adler32_combine(uint32_t adler_a, uint32_t adler_b, uint64_t len_b);

#define SFH_NSTAGES 5 /* 0 lz77 match+parse, 1 code plan, 2 offset scan, 3 emit, 4 checksum + wrapper (container modes) */

/* on != 0: bracket every kernel launch with HIP events on the launch stream */
void sfh_set_profiling(sfh_ctx* ctx, int on);
/* after the stream has been synchronised: milliseconds per stage of the last call */
int sfh_last_stage_ms(sfh_ctx* ctx, float ms[SFH_NSTAGES]);
const char* sfh_stage_name(int stage);

/* ---- stage inspection for parity tests: copies device scratch of the last call ---- */
enum sfh_debug_what {
  SFH_DBG_NTOK = 0,   /* uint32 per chunk */
  SFH_DBG_TOKENS = 1, /* decoder: uint32[32768] per segment, first ntok valid */
  SFH_DBG_ITEMS = 8,  /* compressor: uint16[32768] per chunk, first nitems valid (literal: 0x8000 | byte; match: 0x8100 | len-3,
                         then dist-1 with bit 15 clear; 0x4000 + region index in bits 9..13 on a parse region's first item) */
  SFH_DBG_NITEMS = 9, /* uint32 per chunk */
  SFH_DBG_HIST = 2,   /* uint32[576] per chunk: ll[0..285], d at [288..317], raw len-3 counts at [320..575] */
  SFH_DBG_PLAN = 3,   /* uint32[4] per chunk: btype, out_bytes, header_bits, body_bits */
  SFH_DBG_LENS = 4,   /* uint8[320] per chunk: ll lens [0..287], d lens [288..319] */
  SFH_DBG_OFFSETS = 5, /* uint64 per chunk */
  SFH_DBG_SUBINDEX = 7, /* uint32[64] per chunk */
  SFH_DBG_SEGINFO = 10, /* decoder: uint32[6] per segment {status, tokens, bit 0 stored segment | bit 1 decoded by the
                           lane-serial kernel, bytes, offset of a stored segment's bytes (u64)} */
  SFH_DBG_STAMPS = 6   /* uint64[2][nchunks][8] (k_lz77, k_plan); only with env SFH_K1_STAMPS=1 at sfh_create
                          (diagnostic k_lz77 build: cycles per phase, never a timing claim) */
};
int sfh_debug_read(sfh_ctx* ctx, int what, void* host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
