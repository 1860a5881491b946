// sf_capi.hip -- the C-ABI of include/starflate_hip.h over the kernels of sf_kernels.hip.
// No torch types, no exceptions across the boundary; a ctx owns its device scratch.
#include "../../include/starflate_hip.h"
#include "sf_device.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <new>
#include <thread>
#include <vector>

struct sfh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // used when the caller passes no stream
  sf::Workspace ws{};
  uint32_t cap_chunks = 0;       // chunks the workspace can hold
  uint32_t cap_dtok = 0;         // segments the decoder's token buffer can hold
  size_t sums_cap = 0;           // bytes of ws.sums
  uint32_t last_chunks = 0;
  uint32_t last_block_bytes = 0; // strip size the last compress call used
  bool index_valid = false;      // ws.offsets holds the index of the last compress call
  uint64_t* d_total = nullptr;   // own result slot for the synchronous entry points
  uint64_t* h_total = nullptr;   // ... and the pinned word it is copied to (a pageable target is staged by the runtime)
  uint32_t* d_value = nullptr;   // result slot of sfh_checksum_device; [2] for the decoder's status
  uint64_t* d_index = nullptr;   // staging for the host-buffer decoder
  size_t d_index_cap = 0;
  uint32_t* d_sub = nullptr;
  size_t d_sub_cap = 0;
  // per-stage events of the last profiled decode call: SFH_INFLATE_NSTAGES + 1 per batch, grown on demand
  std::vector<hipEvent_t> ev_inf;
  uint32_t ev_inf_batches = 0;
  bool ev_inf_valid = false;
  size_t last_dtok_bytes = 0;    // bytes of decoder token scratch the last decode call used (bounded: one batch)
  uint8_t* d_in = nullptr;       // staging for the host-buffer entry points (capacities in bytes)
  uint8_t* d_out = nullptr;
  size_t d_in_cap = 0, d_out_cap = 0;
  int profiling = 0;
  int k1_stamps = 0;  // SFH_K1_STAMPS=1: diagnostic k_lz77 build with s_memtime stamps
  uint32_t batch_chunks = sf::kBatchChunks;  // SFH_BATCH_CHUNKS=<n>: smaller batches (tests of the batch loop)
  // per-kernel events of the last profiled call: kEvPerBatch per batch (before k_lz77, k_plan, k_scan, k_emit and after
  // k_emit), then one after the container kernels; grown on demand, reused by later calls
  static constexpr int kEvPerBatch = 5;
  std::vector<hipEvent_t> ev;
  uint32_t ev_batches = 0;       // batches the last profiled call recorded
  bool ev_valid = false;
  // host-buffer path (sfh_compress): copies of one batch run beside the kernels of its neighbours
  static constexpr int kPipe = 4;
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[kPipe] = {}, ev_batch[kPipe] = {};
  uint64_t* h_tot = nullptr;     // pinned: the stream's end after each batch in flight
  hipEvent_t ev_done = nullptr;  // end of the last call's device work: the next call, on any stream, starts behind it
  bool busy = false;             // (the device scratch is shared by all calls on this ctx)
  hipStream_t last_stream = nullptr;
  uint64_t* d_sizes = nullptr;   // sfh_gather_streams: the ranks' sizes on the device and in pinned memory
  uint64_t* h_sizes = nullptr;
  int sizes_cap = 0;
  int order_ok[2] = {0, 0};      // sfh_lds_order_check per op (0 exchange: chains, 1 masked-or: recent): 0 not run, 1 holds, -1 does not
  int force_order_fail = 0;      // SFH_FORCE_ORDER_FAIL=1: the library's own check reports failure (tests)
  int plan_fused = 0;            // SFH_PLAN_FUSED=1: k_plan as one launch (the rounds 1-5 kernel; A/B)
  int inflate_serial = 0;        // SFH_INFLATE_SERIAL=1: index-only streams through the lane-serial kernel alone (tests, A/B)
  char err[256] = {0};
};

namespace {

int fail(sfh_ctx* c, int code, const char* what, hipError_t e) {
  if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
  return code;
}

#define SF_HIP(call, what)                                   \
  do {                                                       \
    hipError_t _e = (call);                                  \
    if (_e != hipSuccess) return fail(ctx, SFH_E_HIP, what, _e); \
  } while (0)

// Grow-only device staging buffer: *p holds at least `bytes` afterwards (*cap_bytes tracks it).
template <class T>
int grow(sfh_ctx* ctx, T** p, size_t* cap_bytes, size_t bytes, const char* what) {
  if (*cap_bytes >= bytes) return SFH_OK;
  (void)hipFree(*p);
  *p = nullptr;
  *cap_bytes = 0;
  const hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return fail(ctx, SFH_E_NOMEM, what, e);
  *cap_bytes = bytes;
  return SFH_OK;
}

void free_ws(sfh_ctx* c) {
  (void)hipFree(c->ws.items);
  (void)hipFree(c->ws.nitems);
  (void)hipFree(c->ws.tokens);
  (void)hipFree(c->ws.ntok);
  (void)hipFree(c->ws.hist);
  (void)hipFree(c->ws.plan);
  (void)hipFree(c->ws.codes);
  (void)hipFree(c->ws.ptree);
  (void)hipFree(c->ws.offsets);
  (void)hipFree(c->ws.stamps);
  (void)hipFree(c->ws.seginfo);
  (void)hipFree(c->ws.rtok);
  (void)hipFree(c->ws.subidx);
  uint32_t* keep = c->ws.sums;  // sized on its own (ensure_sums)
  c->ws = sf::Workspace{};
  c->ws.sums = keep;
  c->cap_chunks = 0;
  c->cap_dtok = 0;
}

// checksum partials: 4 bytes per chunk, needed without the rest of the workspace by sfh_checksum_device
int ensure_sums(sfh_ctx* ctx, uint32_t nchunks) {
  return grow(ctx, &ctx->ws.sums, &ctx->sums_cap, (size_t)nchunks * sizeof(uint32_t), "checksum scratch");
}

// The per-batch arrays hold min(nchunks, kBatchChunks) chunks, the index arrays (offsets, sub-index, segment records)
// every chunk of the call.
int ensure_ws(sfh_ctx* ctx, uint32_t nchunks) {
  if (nchunks <= ctx->cap_chunks) return SFH_OK;
  free_ws(ctx);
  // (a strip larger than a batch is its own batch: kMaxStrip / kChunk = 512 chunks at most)
  const size_t nc = nchunks, nb = std::min<uint32_t>(nchunks, std::max<uint32_t>(ctx->batch_chunks, sf::kMaxStrip / sf::kChunk));
  hipError_t e;
  if ((e = hipMalloc(&ctx->ws.items, nb * sf::kChunk * sizeof(uint16_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.nitems, nb * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.ntok, nb * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.hist, nb * sf::kHistStride * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.plan, nb * sizeof(sf::ChunkPlan))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.codes, nb * sizeof(sf::ChunkCodes))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.ptree, nb * sizeof(sf::PlanTree))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.rtok, nb * sf::kSubRegions * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.offsets, (nc + 1) * sizeof(uint64_t))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.seginfo, nc * sizeof(sf::SegInfo))) != hipSuccess ||
      (e = hipMalloc(&ctx->ws.subidx, nc * 2 * sf::kSubRegions * sizeof(uint32_t))) != hipSuccess) {
    free_ws(ctx);
    return fail(ctx, SFH_E_NOMEM, "workspace hipMalloc", e);
  }
  if (ctx->k1_stamps && (e = hipMalloc(&ctx->ws.stamps, nb * 16 * sizeof(uint64_t))) != hipSuccess) {
    free_ws(ctx);
    return fail(ctx, SFH_E_NOMEM, "stamps hipMalloc", e);
  }
  ctx->cap_chunks = nchunks;
  return SFH_OK;
}

// the decoder's token buffer (4 bytes per output byte) is only allocated once a decode call needs it
int ensure_dtok(sfh_ctx* ctx, uint32_t nseg) {
  if (nseg <= ctx->cap_dtok) return SFH_OK;
  (void)hipFree(ctx->ws.tokens);
  ctx->ws.tokens = nullptr;
  ctx->cap_dtok = 0;
  const hipError_t e = hipMalloc(&ctx->ws.tokens, (size_t)nseg * sf::kChunk * sizeof(uint32_t));
  if (e != hipSuccess) return fail(ctx, SFH_E_NOMEM, "decoder token buffer hipMalloc", e);
  ctx->cap_dtok = nseg;
  return SFH_OK;
}

uint32_t chunks_of(size_t n) { return n ? (uint32_t)((n + sf::kChunk - 1) / sf::kChunk) : 1u; }

// block_bytes = 0: SFH_DEFAULT_BLOCK_BYTES -- larger (SFH_LARGE_BLOCK_BYTES, SFH_CHAIN_BLOCK_BYTES with a chain effort)
// while the input still fills the device four times over with such strips (a strip starts with an empty window: fewer
// starts, a better ratio, the same time per byte), smaller while it has fewer than 256 strips.  A function of n and the
// effort alone, so the stream is too (the encoder specification states the same rule)
uint32_t resolve_block_bytes(uint32_t block_bytes, size_t n, uint32_t effort = SFH_EFFORT_DEFAULT) {
  if (block_bytes) return block_bytes;
  const bool chain = effort >= SFH_EFFORT_BEST && effort <= SFH_EFFORT_EXTREME;
  uint32_t b = chain ? SFH_CHAIN_BLOCK_BYTES : SFH_LARGE_BLOCK_BYTES;
  while (b > SFH_DEFAULT_BLOCK_BYTES && n / b < (chain ? 1024u : 2048u)) b >>= 1;
  while (b > sf::kChunk && n / b < 256) b >>= 1;
  return b;
}

// Calls on one ctx share its device scratch: a call enqueued on another stream than the previous one first waits
// (on the device) for that one's work.
int order_behind_last_call(sfh_ctx* ctx, hipStream_t s) {
  if (ctx->busy && ctx->last_stream != s) SF_HIP(hipStreamWaitEvent(s, ctx->ev_done, 0), "wait for the previous call");
  return SFH_OK;
}
int mark_call_end(sfh_ctx* ctx, hipStream_t s) {
  SF_HIP(hipEventRecord(ctx->ev_done, s), "event");
  ctx->busy = true;
  ctx->last_stream = s;
  return SFH_OK;
}

int check_opt(const sfh_options* o) {
  if (!o) return 0;
  if (o->strategy > SFH_DYNAMIC || o->final_stream > 1 || o->lazy > 3 || o->no_stored_fast_path > 1) return -1;
  if (o->container > SFH_GZIP || (o->container && !o->final_stream)) return -1;  // a non-final shard has no trailer
  if (o->block_bytes % sf::kChunk || o->block_bytes > sf::kMaxStrip) return -1;
  if (o->effort > SFH_EFFORT_RECENT_ALL || o->chain_depth > 255) return -1;
  if (o->chain_depth && (o->effort < SFH_EFFORT_BEST || o->effort > SFH_EFFORT_EXTREME)) return -1;
  return 0;
}

// The returning LDS atomic of the exact-recency match finders (op 0: chains, op 1: recent) executes a wave's lanes in
// ascending order on this device?  Checked once per context IN sfh_create, on the context's own stream (64 workgroups x 4
// steps x 5 densities: 1.2 M positions, well under a millisecond of kernels), so that no compress call -- an asynchronous
// one on the caller's stream, possibly under capture -- ever launches or waits for it; ensure_order() reads the verdict
// (and runs the check itself only if sfh_create could not).
int check_order(sfh_ctx* ctx, int op, uint64_t* bad, uint64_t* checked, uint32_t blocks, uint32_t iters) {
  SF_HIP(sf::run_lds_order_check(op, blocks, iters, ctx->d_value, ctx->stream), "lds order check");
  uint32_t r[2] = {0, 0};
  SF_HIP(hipMemcpyAsync(r, ctx->d_value, sizeof r, hipMemcpyDeviceToHost, ctx->stream), "lds order check result");
  SF_HIP(hipStreamSynchronize(ctx->stream), "lds order check sync");
  *bad = r[0];
  *checked = r[1];
  return SFH_OK;
}
int ensure_order(sfh_ctx* ctx, int op) {
  if (ctx->order_ok[op] == 0) {
    uint64_t bad = 0, checked = 0;
    const int rc = check_order(ctx, op, &bad, &checked, 64, 4);
    if (rc) return rc;
    ctx->order_ok[op] = (bad == 0 && checked != 0 && !ctx->force_order_fail) ? 1 : -1;
  }
  if (ctx->order_ok[op] < 0) {
    snprintf(ctx->err, sizeof ctx->err, "%s does not execute a wave's lanes in ascending order on this device%s: the %s need it",
             op ? "ds_mskor_rtn_b32" : "ds_wrxchg_rtn_b32", ctx->force_order_fail ? " (SFH_FORCE_ORDER_FAIL=1)" : "",
             op ? "effort SFH_EFFORT_RECENT" : "chain efforts (SFH_EFFORT_BEST / _ULTRA / _EXTREME)");
    return SFH_E_UNSUPPORTED;
  }
  return SFH_OK;
}

// Host buffers of sfh_compress: with them the batch loop also moves the data -- batch b's input goes up on one copy
// stream while batch b-1 is in the kernels and batch b-2's stream bytes go down on another.
struct HostPipe {
  const uint8_t* src;
  uint8_t* dst;
  size_t cap;
  size_t copied = 0;   // stream bytes already on their way to dst
  bool overflow = false;
};
constexpr uint32_t kPipeBatchChunks = 2048;  // 64 MiB of input per batch on the host-buffer path

int enqueue(sfh_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t cap, uint64_t* d_out_n,
            const sfh_options* opt, hipStream_t s, HostPipe* pipe = nullptr) {
  if (!ctx || (!d_src && n) || !d_dst || !d_out_n || check_opt(opt)) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  if (((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 3)) return fail(ctx, SFH_E_INVALID_ARG, "device pointer alignment (src 16, dst 4)", hipSuccess);
  if (cap < sfh_compress_bound(n, 0)) return fail(ctx, SFH_E_DST_TOO_SMALL, "cap < sfh_compress_bound(n)", hipSuccess);
  if (n > ((size_t)1 << 44)) return fail(ctx, SFH_E_INVALID_ARG, "input too large", hipSuccess);
  sfh_options o;
  if (opt) o = *opt; else sfh_default_options(&o);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  (void)hipGetLastError();  // launches are checked with hipGetLastError(): drop whatever an earlier caller on this thread left
  const uint32_t nchunks = chunks_of(n);
  int rc = ensure_ws(ctx, nchunks);
  if (!rc && o.container) rc = ensure_sums(ctx, nchunks);
  if (rc) return rc;
  ctx->last_chunks = nchunks;
  // effort -> what the match kernel does: {both levels, near candidate, even positions only, second table, chain depth, exact recency}
  const uint32_t ef = o.effort;
  const bool ef_chain = ef >= SFH_EFFORT_BEST && ef <= SFH_EFFORT_EXTREME, ef_recent = ef == SFH_EFFORT_RECENT || ef == SFH_EFFORT_RECENT_ALL;
  const bool ef_all = ef == SFH_EFFORT_THOROUGH || ef == SFH_EFFORT_MAX || ef_chain || ef == SFH_EFFORT_RECENT_ALL;  // every position searched
  const sf::Options ko{o.strategy, o.final_stream, o.lazy, o.no_stored_fast_path ? 0u : (o.strategy == 0 ? 2u : 1u),
                       resolve_block_bytes(o.block_bytes, n, o.effort),
                       (ef == SFH_EFFORT_FAST || ef == SFH_EFFORT_FASTEST) ? 0u : 1u,
                       ef == SFH_EFFORT_FASTEST ? 0u : 1u, ef_all ? 0u : 1u,
                       ef == SFH_EFFORT_MAX ? 1u : 0u,
                       !ef_chain ? 0u : o.chain_depth ? o.chain_depth
                       : ef == SFH_EFFORT_BEST ? 8u : ef == SFH_EFFORT_ULTRA ? 16u : 32u,
                       ef_recent ? 1u : 0u, ctx->plan_fused ? 1u : 0u};
  if ((ko.chain_depth || ko.recent) && (rc = ensure_order(ctx, ko.recent ? 1 : 0)) != SFH_OK) return rc;
  ctx->last_block_bytes = ko.strip_bytes;
  const bool prof = ctx->profiling != 0;
  if ((rc = order_behind_last_call(ctx, s)) != SFH_OK) return rc;
  // Batches of whole strips, at most kBatchChunks chunks each, one after the other on the stream: strips are coded
  // independently, so the stream is the same as from one launch over everything.  (Per-kernel events: the first batch.)
  const uint32_t per_strip = ko.strip_bytes / sf::kChunk;
  const uint32_t want = pipe ? std::min(ctx->batch_chunks, kPipeBatchChunks) : ctx->batch_chunks;
  const uint32_t batch = std::max(per_strip, want / per_strip * per_strip);
  const uint32_t nbatches = (nchunks + batch - 1) / batch;
  ctx->ev_valid = false;
  if (prof) {
    const size_t need = (size_t)nbatches * sfh_ctx::kEvPerBatch + 1;
    while (ctx->ev.size() < need) {
      hipEvent_t e = nullptr;
      SF_HIP(hipEventCreate(&e), "event");
      ctx->ev.push_back(e);
    }
  }
  const size_t hdr = sf::wrapper_header_bytes(o.container);
  if (pipe) pipe->copied = hdr;  // the wrapper header is written last (k_wrap) and copied last
  // the stream bytes of batch `b` (its end is in h_tot once ev_batch fires) go down while later batches run
  auto drain = [&](uint32_t b) -> int {
    SF_HIP(hipEventSynchronize(ctx->ev_batch[b % sfh_ctx::kPipe]), "wait for a batch");
    const size_t end = (size_t)ctx->h_tot[b % sfh_ctx::kPipe];
    if (end > pipe->cap) { pipe->overflow = true; return SFH_OK; }
    if (end > pipe->copied)
      SF_HIP(hipMemcpyAsync(pipe->dst + pipe->copied, (const uint8_t*)d_dst + pipe->copied, end - pipe->copied,
                            hipMemcpyDeviceToHost, ctx->s_out), "D2H");
    pipe->copied = end;
    return SFH_OK;
  };
  uint32_t bi = 0;
  for (uint32_t c0 = 0; c0 < nchunks; c0 += batch, ++bi) {
    const uint32_t nb = std::min(batch, nchunks - c0);
    const bool first = c0 == 0, last = c0 + nb == nchunks;
    const uint8_t* bsrc = (const uint8_t*)d_src + (size_t)c0 * sf::kChunk;
    const size_t bn = std::min((size_t)nb * sf::kChunk, n - (size_t)c0 * sf::kChunk);
    if (pipe && bn) {
      SF_HIP(hipMemcpyAsync(const_cast<uint8_t*>(bsrc), pipe->src + (size_t)c0 * sf::kChunk, bn, hipMemcpyHostToDevice, ctx->s_in), "H2D");
      SF_HIP(hipEventRecord(ctx->ev_in[bi % sfh_ctx::kPipe], ctx->s_in), "event");
      SF_HIP(hipStreamWaitEvent(s, ctx->ev_in[bi % sfh_ctx::kPipe], 0), "wait for the input");
    }
    sf::Workspace w = ctx->ws;  // this batch's view: index arrays advance, batch arrays start over
    w.offsets += c0;
    w.subidx += (size_t)c0 * 2 * sf::kSubRegions;
    sf::Options bo = ko;
    bo.final_stream = last ? ko.final_stream : 0u;
    // every batch has its own events (recorded behind the wait for its input, so a kernel's time excludes the copy)
    hipEvent_t* ev = prof ? &ctx->ev[(size_t)bi * sfh_ctx::kEvPerBatch] : nullptr;
    if (ev) SF_HIP(hipEventRecord(ev[0], s), "event");
    SF_HIP(sf::launch_lz77(bsrc, bn, nb, w, bo, s), "launch k_lz77");
    if (ev) SF_HIP(hipEventRecord(ev[1], s), "event");
    SF_HIP(sf::launch_plan(bn, nb, w, bo, s), "launch k_plan");
    if (ev) SF_HIP(hipEventRecord(ev[2], s), "event");
    SF_HIP(sf::launch_scan(nb, w, sf::wrapper_header_bytes(o.container), !first, d_out_n, s), "launch k_scan");
    if (ev) SF_HIP(hipEventRecord(ev[3], s), "event");
    SF_HIP(sf::launch_emit(bsrc, bn, nb, w, (uint8_t*)d_dst, s), "launch k_emit");
    if (ev) SF_HIP(hipEventRecord(ev[4], s), "event");
    if (pipe) {
      if (bi >= 1 && (rc = drain(bi - 1)) != SFH_OK) return rc;  // (its slot is free again before batch bi + kPipe - 1 needs it)
      SF_HIP(hipMemcpyAsync(&ctx->h_tot[bi % sfh_ctx::kPipe], d_out_n, sizeof(uint64_t), hipMemcpyDeviceToHost, s), "copy size");
      SF_HIP(hipEventRecord(ctx->ev_batch[bi % sfh_ctx::kPipe], s), "event");
    }
  }
  if (pipe && (rc = drain(bi - 1)) != SFH_OK) return rc;
  if (o.container) {
    SF_HIP(sf::launch_checksum((const uint8_t*)d_src, n, nchunks, o.container, ctx->ws.sums, s), "launch k_checksum");
    SF_HIP(sf::launch_wrap(ctx->ws.sums, nchunks, n, o.container, (uint8_t*)d_dst, d_out_n, nullptr, s), "launch k_wrap");
  }
  if (prof) {
    SF_HIP(hipEventRecord(ctx->ev[(size_t)nbatches * sfh_ctx::kEvPerBatch], s), "event");
    ctx->ev_batches = nbatches;
  }
  ctx->ev_valid = prof;
  ctx->index_valid = true;
  return mark_call_end(ctx, s);
}

}  // namespace

// ---- one process per GPU: concatenation over RCCL (bound at run time: a host without RCCL still loads the library) ----
namespace {
struct Rccl {
  // the subset of rccl.h this file calls (ncclResult_t is an int, ncclComm_t an opaque pointer, ncclUint8 = 1, ncclUint64 = 5)
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(const void*, int*) = nullptr;
  int (*CommUserRank)(const void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  char why[160] = {0};
};
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    // the copy that is in the process already wins (a torch process has loaded its own librccl.so): two RCCLs, like two
    // HIP runtimes, must not serve one communicator
    const char* env = getenv("SFH_RCCL_LIB");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      snprintf(x.why, sizeof x.why, "librccl.so not found (%s)", dlerror());
      return x;
    }
    auto sym = [&](const char* s) { return dlsym(h, s); };
    x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
    x.Send = (decltype(x.Send))sym("ncclSend");
    x.Recv = (decltype(x.Recv))sym("ncclRecv");
    x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
    x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
    x.CommCount = (decltype(x.CommCount))sym("ncclCommCount");
    x.CommUserRank = (decltype(x.CommUserRank))sym("ncclCommUserRank");
    x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
    x.ok = x.AllGather && x.Send && x.Recv && x.GroupStart && x.GroupEnd && x.CommCount && x.CommUserRank;
    if (!x.ok) snprintf(x.why, sizeof x.why, "librccl.so lacks an ncclAllGather / ncclSend / ncclRecv / ncclGroup* symbol");
    return x;
  }();
  return r;
}
int comm_fail(sfh_ctx* ctx, const char* what, int code) {
  const Rccl& R = rccl();
  if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, R.GetErrorString ? R.GetErrorString(code) : "RCCL error");
  return SFH_E_COMM;
}
constexpr int kNcclUint8 = 1, kNcclUint64 = 5;
}  // namespace


extern "C" {

void sfh_default_options(sfh_options* o) {
  memset(o, 0, sizeof *o);
  o->strategy = SFH_AUTO;
  o->final_stream = 1;
  o->lazy = 3;
}

int sfh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sfh_get_device_props(int device, sfh_device_props* out) {
  if (!out) return SFH_E_INVALID_ARG;
  memset(out, 0, sizeof *out);
  const int n = sfh_device_count();
  if (n <= 0 || device < 0 || device >= n) return SFH_E_NO_DEVICE;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return SFH_E_HIP;
  snprintf(out->name, sizeof out->name, "%s", p.name);
  snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
  out->compute_units = (uint32_t)p.multiProcessorCount;
  out->lds_bytes_per_cu = (uint32_t)p.maxSharedMemoryPerMultiProcessor;
  out->l2_bytes = (uint32_t)p.l2CacheSize;
  out->memory_clock_khz = (uint32_t)p.memoryClockRate;
  out->memory_bus_bits = (uint32_t)p.memoryBusWidth;
  out->clock_khz = (uint32_t)p.clockRate;
  out->total_memory = (uint64_t)p.totalGlobalMem;
  return SFH_OK;
}

int sfh_create(sfh_ctx** out, int device) {
  if (!out) return SFH_E_INVALID_ARG;
  *out = nullptr;
  int n = sfh_device_count();
  if (n <= 0 || device < 0 || device >= n) return SFH_E_NO_DEVICE;
  sfh_ctx* ctx = new (std::nothrow) sfh_ctx();
  if (!ctx) return SFH_E_NOMEM;
  ctx->device = device;
  {
    const char* e = getenv("SFH_K1_STAMPS");
    ctx->k1_stamps = (e && e[0] == '1');
    const char* b = getenv("SFH_BATCH_CHUNKS");
    if (b && atoi(b) > 0) ctx->batch_chunks = std::min<uint32_t>((uint32_t)atoi(b), sf::kBatchChunks);
    const char* f = getenv("SFH_FORCE_ORDER_FAIL");
    ctx->force_order_fail = (f && f[0] == '1');
    const char* pf = getenv("SFH_PLAN_FUSED");
    ctx->plan_fused = (pf && pf[0] == '1');
    const char* q = getenv("SFH_INFLATE_SERIAL");
    ctx->inflate_serial = (q && q[0] == '1');
  }
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
      (e = hipMalloc(&ctx->d_total, sizeof(uint64_t))) != hipSuccess ||
      (e = hipHostMalloc((void**)&ctx->h_total, sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess ||
      (e = hipMalloc(&ctx->d_value, 2 * sizeof(uint32_t))) != hipSuccess || (e = sf::init_kernels()) != hipSuccess ||
      (e = sf::init_inflate_kernels()) != hipSuccess) {
    sfh_destroy(ctx);
    return SFH_E_HIP;
  }
  if (hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess) {
    sfh_destroy(ctx);
    return SFH_E_HIP;
  }
  for (int op = 0; op < 2; ++op) {  // the LDS ordering the chain and recent efforts rest on: settled here, once (see ensure_order)
    uint64_t bad = 0, checked = 0;
    if (check_order(ctx, op, &bad, &checked, 64, 4) == SFH_OK)
      ctx->order_ok[op] = (bad == 0 && checked != 0 && !ctx->force_order_fail) ? 1 : -1;
    ctx->err[0] = 0;  // (a check that could not run is run again by the first call that needs it, which then reports)
  }
  *out = ctx;
  return SFH_OK;
}

void sfh_destroy(sfh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_ws(ctx);
  (void)hipFree(ctx->ws.sums);
  (void)hipFree(ctx->d_total);
  if (ctx->h_total) (void)hipHostFree(ctx->h_total);
  (void)hipFree(ctx->d_value);
  (void)hipFree(ctx->d_index);
  (void)hipFree(ctx->d_sub);
  for (hipEvent_t e : ctx->ev_inf) (void)hipEventDestroy(e);
  (void)hipFree(ctx->d_in);
  (void)hipFree(ctx->d_out);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
  for (int k = 0; k < sfh_ctx::kPipe; ++k) {
    if (ctx->ev_in[k]) (void)hipEventDestroy(ctx->ev_in[k]);
    if (ctx->ev_batch[k]) (void)hipEventDestroy(ctx->ev_batch[k]);
  }
  if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
  if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
  if (ctx->h_tot) (void)hipHostFree(ctx->h_tot);
  (void)hipFree(ctx->d_sizes);
  if (ctx->h_sizes) (void)hipHostFree(ctx->h_sizes);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* sfh_last_error(const sfh_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int sfh_lds_order_check(sfh_ctx* ctx, uint32_t op, uint32_t blocks, uint32_t iters, uint64_t* mismatches, uint64_t* checked) {
  // (the kernel counts mismatches and checked positions with 32-bit atomics: blocks * iters * 1024 positions * 5 densities must fit,
  // and a launch at the bound still ends within seconds)
  if (!ctx || op > 1 || !blocks || blocks > 4096 || !iters || iters > 128 || !mismatches || !checked)
    return fail(ctx, SFH_E_INVALID_ARG, "argument (blocks 1..4096, iters 1..128)", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  return check_order(ctx, (int)op, mismatches, checked, blocks, iters);
}


size_t sfh_compress_bound(size_t n, uint32_t block_bytes) {
  // per 32 KiB DEFLATE block: fixed-Huffman worst case (9 bits per literal) + headers + alignment block.  A strip is a
  // whole number of such blocks whatever block_bytes is, so a VALID block_bytes does not change the bound; one that
  // sfh_compress would reject (not a multiple of 32 KiB, above 16 MiB) gives 0: no buffer size makes that call succeed
  if (block_bytes && (block_bytes % sf::kChunk || block_bytes > sf::kMaxStrip)) return 0;
  const size_t nchunks = n ? (n + sf::kChunk - 1) / sf::kChunk : 1;
  return nchunks * (size_t)(sf::kChunk + sf::kChunk / 8 + 640);
}

int sfh_compress_device_async(sfh_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t cap,
                              uint64_t* d_out_n, const sfh_options* opt, void* stream) {
  if (!ctx) return SFH_E_INVALID_ARG;
  return enqueue(ctx, d_src, n, d_dst, cap, d_out_n, opt, stream ? (hipStream_t)stream : ctx->stream);
}

int sfh_compress_device(sfh_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t cap,
                        size_t* out_n, const sfh_options* opt, void* stream) {
  if (!ctx || !out_n) return SFH_E_INVALID_ARG;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int rc = enqueue(ctx, d_src, n, d_dst, cap, ctx->d_total, opt, s);
  if (rc) return rc;
  SF_HIP(hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, s), "copy size");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  *out_n = (size_t)*ctx->h_total;
  return SFH_OK;
}

int sfh_compress(sfh_ctx* ctx, const void* src, size_t n, void* dst, size_t cap, size_t* out_n,
                 const sfh_options* opt) {
  if (!ctx || (!src && n) || !dst || !out_n) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  const size_t bound = sfh_compress_bound(n, 0);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  int rc = grow(ctx, &ctx->d_in, &ctx->d_in_cap, n ? n : 16, "input staging");
  if (!rc) rc = grow(ctx, &ctx->d_out, &ctx->d_out_cap, bound, "output staging");
  if (rc) return rc;
  if (!ctx->s_in) {  // the copy streams, their events and the pinned size slots: on first use
    SF_HIP(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking), "stream");
    SF_HIP(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking), "stream");
    for (int k = 0; k < sfh_ctx::kPipe; ++k) {
      SF_HIP(hipEventCreateWithFlags(&ctx->ev_in[k], hipEventDisableTiming), "event");
      SF_HIP(hipEventCreateWithFlags(&ctx->ev_batch[k], hipEventDisableTiming), "event");
    }
    SF_HIP(hipHostMalloc((void**)&ctx->h_tot, sfh_ctx::kPipe * sizeof(uint64_t), hipHostMallocDefault), "pinned slots");
  }
  hipStream_t s = ctx->stream;
  // the staging buffers may still be read by copies of the previous call: they were all waited for below
  HostPipe pipe{(const uint8_t*)src, (uint8_t*)dst, cap};
  rc = enqueue(ctx, ctx->d_in, n, ctx->d_out, bound, ctx->d_total, opt, s, &pipe);
  if (rc) {
    (void)hipStreamSynchronize(ctx->s_in);
    (void)hipStreamSynchronize(s);
    (void)hipStreamSynchronize(ctx->s_out);
    return rc;
  }
  uint64_t total = 0;
  SF_HIP(hipMemcpyAsync(&total, ctx->d_total, sizeof total, hipMemcpyDeviceToHost, s), "copy size");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  if (pipe.overflow || total > cap) {
    (void)hipStreamSynchronize(ctx->s_out);
    return fail(ctx, SFH_E_DST_TOO_SMALL, "dst capacity below stream size", hipSuccess);
  }
  // a wrapped stream: the header in front and the trailer behind the raw bytes came last (k_wrap)
  const sfh_options* o = opt;
  const size_t hdr = (o && o->container) ? sf::wrapper_header_bytes(o->container) : 0;
  if (hdr) SF_HIP(hipMemcpyAsync(dst, ctx->d_out, hdr, hipMemcpyDeviceToHost, ctx->s_out), "D2H header");
  if (total > pipe.copied)
    SF_HIP(hipMemcpyAsync((uint8_t*)dst + pipe.copied, ctx->d_out + pipe.copied, total - pipe.copied, hipMemcpyDeviceToHost, ctx->s_out), "D2H trailer");
  SF_HIP(hipStreamSynchronize(ctx->s_out), "stream sync");
  *out_n = (size_t)total;
  return SFH_OK;
}

uint32_t sfh_last_block_bytes(const sfh_ctx* ctx) { return (ctx && ctx->index_valid) ? ctx->last_block_bytes : 0u; }

size_t sfh_index_entries(const sfh_ctx* ctx) { return (ctx && ctx->index_valid) ? (size_t)ctx->last_chunks + 1 : 0; }

int sfh_copy_index(sfh_ctx* ctx, uint64_t* dst, size_t entries, int dst_on_device, void* stream) {
  if (!ctx || !dst || !ctx->index_valid || entries != (size_t)ctx->last_chunks + 1)
    return fail(ctx, SFH_E_INVALID_ARG, "index: no compress call yet, or entries != segments + 1", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  if (int rc = order_behind_last_call(ctx, s)) return rc;  // the index is written by the last call's k_scan, maybe on another stream
  SF_HIP(hipMemcpyAsync(dst, ctx->ws.offsets, entries * sizeof(uint64_t),
                        dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s), "copy index");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  return SFH_OK;
}

int sfh_copy_subindex(sfh_ctx* ctx, uint32_t* dst, size_t words, int dst_on_device, void* stream) {
  if (!ctx || !dst || !ctx->index_valid || words != (size_t)ctx->last_chunks * SFH_SUBINDEX_WORDS)
    return fail(ctx, SFH_E_INVALID_ARG, "sub-index: no compress call yet, or words != segments * 64", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  if (int rc = order_behind_last_call(ctx, s)) return rc;  // written by the last call's k_emit
  SF_HIP(hipMemcpyAsync(dst, ctx->ws.subidx, words * sizeof(uint32_t),
                        dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s), "copy sub-index");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  return SFH_OK;
}

int sfh_decompress_device(sfh_ctx* ctx, const void* d_src, size_t src_n, const uint64_t* d_index,
                          const uint32_t* d_subindex, size_t nseg, void* d_dst, size_t dst_n, uint32_t block_bytes,
                          uint32_t* status, void* stream) {
  if (!ctx || !d_src || !d_index || !status || (!d_dst && dst_n)) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  if (((uintptr_t)d_src & 3) || ((uintptr_t)d_index & 7) || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_subindex & 3))
    return fail(ctx, SFH_E_INVALID_ARG, "device pointer alignment (src 4, index 8, dst 16, sub-index 4)", hipSuccess);
  if (dst_n > ((size_t)1 << 44) || nseg != (size_t)chunks_of(dst_n))
    return fail(ctx, SFH_E_INVALID_ARG, "nseg != ceil(dst_n / 32768)", hipSuccess);
  if (block_bytes % sf::kChunk || block_bytes > sf::kMaxStrip)
    return fail(ctx, SFH_E_INVALID_ARG, "block_bytes: a multiple of 32768 up to 16 MiB (0 = 32768)", hipSuccess);
  const uint32_t sps = block_bytes ? block_bytes / sf::kChunk : 1u;  // segments per strip
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  (void)hipGetLastError();  // see enqueue()
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // Batches of whole strips, like the compressor: the token scratch (4 bytes per output byte) holds ONE batch of at most
  // kBatchChunks segments -- 4 GiB for any size of call -- and the two kernels alternate on the stream, batch after batch.
  // Strips decode independently (a match never reaches before its strip, /root/reference/src/decompress.cpp:178 within one), the
  // segment records and the statuses cover the whole call, so the result -- bytes, first failing segment, its status -- is
  // that of one pass over everything.
  const uint32_t batch = std::max(sps, ctx->batch_chunks / sps * sps);  // (a strip larger than a batch is its own batch)
  const uint32_t nbatches = (uint32_t)((nseg + batch - 1) / batch);
  int rc = ensure_ws(ctx, (uint32_t)nseg);
  if (!rc) rc = ensure_dtok(ctx, (uint32_t)std::min<size_t>(nseg, batch));
  if (rc) return rc;
  ctx->last_dtok_bytes = std::min<size_t>(nseg, batch) * sf::kChunk * sizeof(uint32_t);
  ctx->index_valid = false;  // the decoder reuses the scratch: what sfh_debug_read returns now belongs to this call
  ctx->last_chunks = (uint32_t)nseg;
  const bool prof = ctx->profiling != 0;
  if ((rc = order_behind_last_call(ctx, s)) != SFH_OK) return rc;
  ctx->ev_inf_valid = false;
  if (prof) {
    const size_t need = (size_t)nbatches * (SFH_INFLATE_NSTAGES + 1);
    while (ctx->ev_inf.size() < need) {
      hipEvent_t e = nullptr;
      SF_HIP(hipEventCreate(&e), "event");
      ctx->ev_inf.push_back(e);
    }
  }
  uint32_t bi = 0;
  for (size_t g0 = 0; g0 < nseg; g0 += batch, ++bi) {
    const uint32_t nb = (uint32_t)std::min<size_t>(batch, nseg - g0);
    const uint64_t bdst_n = std::min<uint64_t>((uint64_t)nb * sf::kChunk, dst_n - (uint64_t)g0 * sf::kChunk);  // this batch's output bytes
    const uint64_t* bindex = d_index + g0;           // (stream offsets are absolute: src stays what it is)
    sf::SegInfo* binfo = ctx->ws.seginfo + g0;
    uint8_t* bdst = (uint8_t*)d_dst + g0 * sf::kChunk;
    hipEvent_t* ev = prof ? &ctx->ev_inf[(size_t)bi * (SFH_INFLATE_NSTAGES + 1)] : nullptr;
    if (ev) SF_HIP(hipEventRecord(ev[0], s), "event");
    if (d_subindex)
      SF_HIP(sf::launch_inflate_tokens_sub((const uint8_t*)d_src, src_n, bindex, d_subindex + g0 * SFH_SUBINDEX_WORDS, nb, bdst_n,
                                           ctx->ws.tokens, binfo, sps, s), "launch k_inflate_tokens_sub");
    else
      SF_HIP(sf::launch_inflate_tokens((const uint8_t*)d_src, src_n, bindex, nb, bdst_n, ctx->ws.tokens,
                                       binfo, sps, !ctx->inflate_serial, s), "launch k_inflate_tokens");
    if (ev) SF_HIP(hipEventRecord(ev[1], s), "event");
    SF_HIP(sf::launch_inflate_bytes((const uint8_t*)d_src, src_n, nb, ctx->ws.tokens, binfo, bdst, sps, s), "launch k_inflate_bytes");
    if (ev) SF_HIP(hipEventRecord(ev[2], s), "event");
  }
  ctx->ev_inf_batches = nbatches;
  ctx->ev_inf_valid = prof;
  SF_HIP(sf::launch_inflate_status(ctx->ws.seginfo, (uint32_t)nseg, ctx->d_value, s), "launch k_inflate_status");
  uint32_t res[2] = {0, 0};
  SF_HIP(hipMemcpyAsync(res, ctx->d_value, sizeof res, hipMemcpyDeviceToHost, s), "copy status");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  *status = res[0];
  if (res[0]) snprintf(ctx->err, sizeof ctx->err, "segment %u: DecompressStatus %u", res[1], res[0]);
  return SFH_OK;
}

int sfh_decompress(sfh_ctx* ctx, const void* src, size_t src_n, const uint64_t* index, const uint32_t* subindex,
                   size_t nseg, void* dst, size_t dst_n, uint32_t block_bytes, uint32_t* status) {
  if (!ctx || !src || !index || !status || (!dst && dst_n)) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  int rc = grow(ctx, &ctx->d_in, &ctx->d_in_cap, src_n ? src_n : 16, "input staging");
  if (!rc) rc = grow(ctx, &ctx->d_out, &ctx->d_out_cap, dst_n ? dst_n : 16, "output staging");
  if (!rc) rc = grow(ctx, &ctx->d_index, &ctx->d_index_cap, (nseg + 1) * sizeof(uint64_t), "index staging");
  if (!rc && subindex) rc = grow(ctx, &ctx->d_sub, &ctx->d_sub_cap, nseg * SFH_SUBINDEX_WORDS * sizeof(uint32_t), "sub-index staging");
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  if (src_n) SF_HIP(hipMemcpyAsync(ctx->d_in, src, src_n, hipMemcpyHostToDevice, s), "H2D");
  SF_HIP(hipMemcpyAsync(ctx->d_index, index, (nseg + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s), "H2D index");
  if (subindex)
    SF_HIP(hipMemcpyAsync(ctx->d_sub, subindex, nseg * SFH_SUBINDEX_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, s), "H2D sub-index");
  rc = sfh_decompress_device(ctx, ctx->d_in, src_n, ctx->d_index, subindex ? ctx->d_sub : nullptr, nseg, ctx->d_out,
                             dst_n, block_bytes, status, s);
  if (rc) return rc;
  if (*status == 0 && dst_n) {
    SF_HIP(hipMemcpyAsync(dst, ctx->d_out, dst_n, hipMemcpyDeviceToHost, s), "D2H");
    SF_HIP(hipStreamSynchronize(s), "stream sync");
  }
  return SFH_OK;
}

size_t sfh_last_decode_scratch_bytes(const sfh_ctx* ctx) { return ctx ? ctx->last_dtok_bytes : 0; }

int sfh_last_inflate_ms(sfh_ctx* ctx, float ms[SFH_INFLATE_NSTAGES]) {
  if (!ctx || !ms || !ctx->ev_inf_valid) return SFH_E_INVALID_ARG;
  for (int k = 0; k < SFH_INFLATE_NSTAGES; ++k) ms[k] = 0.f;
  for (uint32_t b = 0; b < ctx->ev_inf_batches; ++b)  // a stage's time over every batch of the call
    for (int k = 0; k < SFH_INFLATE_NSTAGES; ++k) {
      float t = 0.f;
      const hipEvent_t* ev = &ctx->ev_inf[(size_t)b * (SFH_INFLATE_NSTAGES + 1)];
      SF_HIP(hipEventElapsedTime(&t, ev[k], ev[k + 1]), "elapsed");
      ms[k] += t;
    }
  return SFH_OK;
}

const char* sfh_inflate_stage_name(int stage) {
  static const char* names[SFH_INFLATE_NSTAGES] = {"k_inflate_tokens", "k_inflate_bytes"};
  return (stage >= 0 && stage < SFH_INFLATE_NSTAGES) ? names[stage] : "";
}

int sfh_checksum_device(sfh_ctx* ctx, const void* d_src, size_t n, uint32_t kind, uint32_t* out, void* stream) {
  if (!ctx || (!d_src && n) || !out || (kind != SFH_ZLIB && kind != SFH_GZIP)) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  if ((uintptr_t)d_src & 15) return fail(ctx, SFH_E_INVALID_ARG, "device pointer alignment (src 16)", hipSuccess);
  if (n > ((size_t)1 << 44)) return fail(ctx, SFH_E_INVALID_ARG, "input too large", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  (void)hipGetLastError();  // see enqueue()
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  const uint32_t nchunks = chunks_of(n);
  int rc = ensure_sums(ctx, nchunks);
  if (!rc) rc = order_behind_last_call(ctx, s);
  if (rc) return rc;
  SF_HIP(sf::launch_checksum((const uint8_t*)d_src, n, nchunks, kind, ctx->ws.sums, s), "launch k_checksum");
  SF_HIP(sf::launch_wrap(ctx->ws.sums, nchunks, n, kind, nullptr, nullptr, ctx->d_value, s), "launch k_wrap");
  SF_HIP(hipMemcpyAsync(out, ctx->d_value, sizeof *out, hipMemcpyDeviceToHost, s), "copy checksum");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  return SFH_OK;
}

uint32_t sfh_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return sf::crc32_combine(crc_a, crc_b, len_b); }
uint32_t sfh_adler32_combine(uint32_t adler_a, uint32_t adler_b, uint64_t len_b) {
  return sf::adler32_combine(adler_a, adler_b, len_b);
}

int sfh_compress_multi(sfh_ctx* const* ctxs, int nctx, const void* src, size_t n, void* dst, size_t cap, size_t* out_n,
                       const sfh_options* opt) {
  if (!ctxs || nctx <= 0 || (!src && n) || !dst || !out_n || check_opt(opt)) return SFH_E_INVALID_ARG;
  for (int i = 0; i < nctx; ++i) {
    if (!ctxs[i]) return SFH_E_INVALID_ARG;
    for (int j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return fail(ctxs[i], SFH_E_INVALID_ARG, "sfh_compress_multi: the same ctx twice", hipSuccess);
  }
  if (cap < sfh_compress_bound(n, 0)) return fail(ctxs[0], SFH_E_DST_TOO_SMALL, "cap < sfh_compress_bound(n)", hipSuccess);
  sfh_options o;
  if (opt) o = *opt; else sfh_default_options(&o);
  // the strip size is fixed once, from the whole input: every shard is a whole number of strips, so the
  // shards' streams are exactly the pieces of the single-call stream
  o.block_bytes = resolve_block_bytes(o.block_bytes, n, o.effort);
  // shards: equal numbers of strips, the tail shards may be empty (they then contribute nothing)
  const size_t cps = o.block_bytes / sf::kChunk;  // chunks per strip
  const size_t nstrips = n ? (n + o.block_bytes - 1) / o.block_bytes : 1;
  const size_t per = ((nstrips + (size_t)nctx - 1) / (size_t)nctx) * cps;
  struct Shard {
    size_t lo = 0, len = 0, bound_off = 0, out = 0;
    uint32_t sum = 0;
    int rc = SFH_OK;
    bool used = false;
  };
  std::vector<Shard> sh((size_t)nctx);
  int last = 0;
  const size_t hdr = sf::wrapper_header_bytes(o.container);
  size_t off = hdr;  // the wrapper header goes in front of the first slice
  for (int i = 0; i < nctx; ++i) {
    Shard& s = sh[(size_t)i];
    s.lo = std::min(n, (size_t)i * per * sf::kChunk);
    const size_t hi = std::min(n, ((size_t)i + 1) * per * sf::kChunk);
    s.len = hi - s.lo;
    s.used = s.len > 0 || i == 0;  // an empty input is one empty stream on the first ctx
    if (s.used) last = i;
    s.bound_off = off;
    if (s.used) off += sfh_compress_bound(s.len, 0);
  }
  // every shard compresses into its own slice of dst (sized by the bound), then the slices are closed up
  std::vector<std::thread> workers;
  for (int i = 0; i < nctx; ++i) {
    if (!sh[(size_t)i].used) continue;
    workers.emplace_back([&, i] {
      Shard& s = sh[(size_t)i];
      sfh_options so = o;
      so.container = SFH_RAW;
      so.final_stream = (i == last) ? o.final_stream : 0u;
      // a stream is shorter than its bound by more than the wrapper (>= 345 bytes of slack per chunk), so the
      // slices, laid out bound after bound behind the header, stay inside cap = sfh_compress_bound(n)
      s.rc = sfh_compress(ctxs[i], (const uint8_t*)src + s.lo, s.len, (uint8_t*)dst + s.bound_off,
                          sfh_compress_bound(s.len, 0), &s.out, &so);
      if (s.rc == SFH_OK && o.container)  // the shard is still staged on the device: checksum it there
        s.rc = sfh_checksum_device(ctxs[i], ctxs[i]->d_in, s.len, o.container, &s.sum, nullptr);
    });
  }
  for (auto& w : workers) w.join();
  for (int i = 0; i < nctx; ++i)
    if (sh[(size_t)i].used && sh[(size_t)i].rc != SFH_OK) return sh[(size_t)i].rc;
  uint8_t* d = (uint8_t*)dst;
  size_t pos = hdr;
  uint32_t sum = 0;
  bool first = true;
  for (int i = 0; i < nctx; ++i) {
    const Shard& s = sh[(size_t)i];
    if (!s.used) continue;
    memmove(d + pos, d + s.bound_off, s.out);  // pos <= bound_off: slices only move down
    pos += s.out;
    if (o.container) {
      sum = first ? s.sum : (o.container == SFH_ZLIB ? sf::adler32_combine(sum, s.sum, s.len) : sf::crc32_combine(sum, s.sum, s.len));
      first = false;
    }
  }
  if (o.container == SFH_ZLIB) {
    d[0] = 0x78;
    d[1] = 0x9C;
    for (int k = 0; k < 4; ++k) d[pos + (size_t)k] = (uint8_t)(sum >> (24 - 8 * k));
    pos += 4;
  } else if (o.container == SFH_GZIP) {
    static const uint8_t h[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF};
    memcpy(d, h, 10);
    const uint32_t isize = (uint32_t)n;
    for (int k = 0; k < 4; ++k) {
      d[pos + (size_t)k] = (uint8_t)(sum >> (8 * k));
      d[pos + 4 + (size_t)k] = (uint8_t)(isize >> (8 * k));
    }
    pos += 8;
  }
  *out_n = pos;
  return SFH_OK;
}

// ---- one process per GPU: concatenation over RCCL (binding: struct Rccl above) ----
int sfh_gather_offsets(const uint64_t* sizes, int nranks, uint64_t base, uint64_t cap, uint64_t* offsets) {
  if (!sizes || !offsets || nranks <= 0) return SFH_E_INVALID_ARG;
  uint64_t at = base;
  for (int r = 0; r < nranks; ++r) {
    offsets[r] = at;
    if (sizes[r] > UINT64_MAX - at) return SFH_E_INVALID_ARG;
    at += sizes[r];
  }
  offsets[nranks] = at;
  return at > cap ? SFH_E_DST_TOO_SMALL : SFH_OK;
}

int sfh_comm_ranks(void* nccl_comm, int* nranks, int* rank) {
  if (!nccl_comm || !nranks || !rank) return SFH_E_INVALID_ARG;
  const Rccl& R = rccl();
  if (!R.ok || R.CommCount(nccl_comm, nranks) != 0 || R.CommUserRank(nccl_comm, rank) != 0) return SFH_E_COMM;
  return SFH_OK;
}

int sfh_gather_streams(sfh_ctx* ctx, void* nccl_comm, int root, const void* d_stream, const uint64_t* d_size, void* d_out,
                       uint64_t base, uint64_t cap, uint64_t* h_sizes, uint64_t* out_end, void* stream) {
  if (!ctx || !nccl_comm || !d_stream || !d_size || !h_sizes || !out_end) return fail(ctx, SFH_E_INVALID_ARG, "argument", hipSuccess);
  const Rccl& R = rccl();
  if (!R.ok) {
    snprintf(ctx->err, sizeof ctx->err, "%s", R.why);
    return SFH_E_COMM;
  }
  int nranks = 0, rank = -1, rc;
  if ((rc = R.CommCount(nccl_comm, &nranks)) != 0) return comm_fail(ctx, "ncclCommCount", rc);
  if ((rc = R.CommUserRank(nccl_comm, &rank)) != 0) return comm_fail(ctx, "ncclCommUserRank", rc);
  if (root < 0 || root >= nranks || (rank == root && !d_out)) return fail(ctx, SFH_E_INVALID_ARG, "root / d_out", hipSuccess);
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // Every rank contributes {size, base, cap, its d_stream, its d_out}: sizes are what is gathered; base and cap must be the
  // same on all ranks, and the root's two addresses let EVERY rank judge the root's own placement -- so every rank reaches
  // the same verdict BEFORE any transfer is posted, and nobody is left waiting in a send whose receive was refused.
  constexpr int kRec = 5;
  if (ctx->sizes_cap < nranks) {
    (void)hipFree(ctx->d_sizes);
    (void)hipHostFree(ctx->h_sizes);
    ctx->d_sizes = ctx->h_sizes = nullptr;
    ctx->sizes_cap = 0;
    SF_HIP(hipMalloc(&ctx->d_sizes, ((size_t)nranks + 1) * kRec * sizeof(uint64_t)), "sizes");
    SF_HIP(hipHostMalloc((void**)&ctx->h_sizes, ((size_t)nranks + 1) * kRec * sizeof(uint64_t), hipHostMallocDefault), "pinned sizes");
    ctx->sizes_cap = nranks;
  }
  uint64_t* const d_mine = ctx->d_sizes + (size_t)nranks * kRec;  // this rank's record, behind the gathered ones
  uint64_t* const h_mine = ctx->h_sizes + (size_t)nranks * kRec;
  h_mine[1] = base;
  h_mine[2] = cap;
  h_mine[3] = (uint64_t)(uintptr_t)d_stream;
  h_mine[4] = (uint64_t)(uintptr_t)d_out;
  // the compressor may have run on another stream of this ctx: its size word is final behind ev_done
  rc = order_behind_last_call(ctx, s);
  if (rc) return rc;
  SF_HIP(hipMemcpyAsync(d_mine + 1, h_mine + 1, (kRec - 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s), "record");
  SF_HIP(hipMemcpyAsync(d_mine, d_size, sizeof(uint64_t), hipMemcpyDeviceToDevice, s), "size word");
  if ((rc = R.AllGather(d_mine, ctx->d_sizes, kRec, kNcclUint64, nccl_comm, s)) != 0) return comm_fail(ctx, "ncclAllGather", rc);
  SF_HIP(hipMemcpyAsync(ctx->h_sizes, ctx->d_sizes, (size_t)nranks * kRec * sizeof(uint64_t), hipMemcpyDeviceToHost, s), "sizes read-back");
  SF_HIP(hipStreamSynchronize(s), "stream sync");
  // (no exception may cross the C boundary: the offsets live in a nothrow allocation)
  struct Free { void operator()(uint64_t* p) const { free(p); } };
  const std::unique_ptr<uint64_t, Free> off_mem((uint64_t*)malloc(((size_t)nranks + 1) * sizeof(uint64_t)));
  if (!off_mem) return fail(ctx, SFH_E_NOMEM, "sfh_gather_streams: offsets", hipSuccess);
  uint64_t* const off = off_mem.get();
  bool agree = true;
  for (int r = 0; r < nranks; ++r) {
    const uint64_t* rec = ctx->h_sizes + (size_t)r * kRec;
    h_sizes[r] = rec[0];
    agree = agree && rec[1] == base && rec[2] == cap;
  }
  if (!agree) return fail(ctx, SFH_E_INVALID_ARG, "sfh_gather_streams: the ranks disagree on base / cap", hipSuccess);  // (on every rank alike)
  rc = sfh_gather_offsets(h_sizes, nranks, base, cap, off);
  *out_end = off[(size_t)nranks];
  if (rc != SFH_OK) return fail(ctx, rc, "sfh_gather_streams: the gathered streams do not fit cap", hipSuccess);
  {
    // the root's own stream: already at its place in d_out, or clear of everything the gather writes -- anything between
    // would be overwritten by a peer's bytes or copied onto itself
    const uint64_t* rr = ctx->h_sizes + (size_t)root * kRec;
    const uint64_t rs = rr[3], ro = rr[4], rn = rr[0];
    const uint64_t w0 = ro + base, w1 = ro + off[(size_t)nranks];
    if (rn && rs != ro + off[(size_t)root] && rs < w1 && rs + rn > w0)
      return fail(ctx, SFH_E_INVALID_ARG, "sfh_gather_streams: the root's d_stream overlaps the gathered range of d_out without being at its own place", hipSuccess);
  }
  if (rank != root) {
    if (h_sizes[rank] && (rc = R.Send(d_stream, (size_t)h_sizes[rank], kNcclUint8, root, nccl_comm, s)) != 0) return comm_fail(ctx, "ncclSend", rc);
    return SFH_OK;
  }
  uint8_t* out = (uint8_t*)d_out;
  if ((rc = R.GroupStart()) != 0) return comm_fail(ctx, "ncclGroupStart", rc);
  for (int r = 0; r < nranks; ++r) {
    if (r == root || !h_sizes[r]) continue;
    if ((rc = R.Recv(out + off[(size_t)r], (size_t)h_sizes[r], kNcclUint8, r, nccl_comm, s)) != 0) {
      (void)R.GroupEnd();
      return comm_fail(ctx, "ncclRecv", rc);
    }
  }
  if ((rc = R.GroupEnd()) != 0) return comm_fail(ctx, "ncclGroupEnd", rc);
  if (h_sizes[root] && (const uint8_t*)d_stream != out + off[(size_t)root])
    SF_HIP(hipMemcpyAsync(out + off[(size_t)root], d_stream, (size_t)h_sizes[root], hipMemcpyDeviceToDevice, s), "own stream");
  return SFH_OK;
}

void sfh_set_profiling(sfh_ctx* ctx, int on) {
  if (ctx) ctx->profiling = on;
}

int sfh_last_stage_ms(sfh_ctx* ctx, float ms[SFH_NSTAGES]) {
  if (!ctx || !ms || !ctx->ev_valid) return SFH_E_INVALID_ARG;
  // summed over every batch of the call (a > 1 GiB device call or any host-buffer call runs several)
  for (int k = 0; k < SFH_NSTAGES; ++k) ms[k] = 0.f;
  for (uint32_t b = 0; b < ctx->ev_batches; ++b) {
    const hipEvent_t* e = &ctx->ev[(size_t)b * sfh_ctx::kEvPerBatch];
    for (int k = 0; k < 4; ++k) {
      float t = 0.f;
      SF_HIP(hipEventElapsedTime(&t, e[k], e[k + 1]), "elapsed");
      ms[k] += t;
    }
  }
  const size_t last = (size_t)ctx->ev_batches * sfh_ctx::kEvPerBatch;
  SF_HIP(hipEventElapsedTime(&ms[4], ctx->ev[last - 1], ctx->ev[last]), "elapsed");  // k_checksum + k_wrap
  return SFH_OK;
}

const char* sfh_stage_name(int stage) {
  static const char* names[SFH_NSTAGES] = {"k_lz77", "k_plan", "k_scan", "k_emit", "k_checksum"};
  return (stage >= 0 && stage < SFH_NSTAGES) ? names[stage] : "";
}

int sfh_debug_read(sfh_ctx* ctx, int what, void* host_dst, size_t bytes) {
  if (!ctx || !host_dst || !ctx->last_chunks) return SFH_E_INVALID_ARG;
  // per-batch arrays hold the last batch of the call (all of it for up to kBatchChunks chunks)
  const size_t nc = std::min<size_t>(ctx->last_chunks, std::max<uint32_t>(ctx->batch_chunks, sf::kMaxStrip / sf::kChunk));
  const size_t nall = ctx->last_chunks;
  if (ctx->busy) SF_HIP(hipEventSynchronize(ctx->ev_done), "wait for the last call");  // whatever stream it ran on
  const void* p = nullptr;
  size_t avail = 0;
  switch (what) {
    case SFH_DBG_NTOK: p = ctx->ws.ntok; avail = nc * 4; break;
    case SFH_DBG_TOKENS: p = ctx->ws.tokens; avail = ctx->cap_dtok >= nc ? nc * sf::kChunk * 4 : 0; break;
    case SFH_DBG_ITEMS: p = ctx->ws.items; avail = nc * sf::kChunk * 2; break;
    case SFH_DBG_NITEMS: p = ctx->ws.nitems; avail = nc * 4; break;
    case SFH_DBG_HIST: p = ctx->ws.hist; avail = nc * sf::kHistStride * 4; break;
    case SFH_DBG_PLAN: p = ctx->ws.plan; avail = nc * sizeof(sf::ChunkPlan); break;
    case SFH_DBG_OFFSETS: p = ctx->ws.offsets; avail = nall * 8; break;
    case SFH_DBG_SUBINDEX: p = ctx->ws.subidx; avail = nall * SFH_SUBINDEX_WORDS * 4; break;
    case SFH_DBG_STAMPS: p = ctx->ws.stamps; avail = p ? nc * 128 : 0; break;
    case SFH_DBG_SEGINFO: p = ctx->ws.seginfo; avail = nall * sizeof(sf::SegInfo); break;
    case SFH_DBG_LENS: {
      if (bytes > nc * 320) return SFH_E_INVALID_ARG;
      SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
      SF_HIP(hipMemcpy2D(host_dst, 320, (const uint8_t*)ctx->ws.codes + offsetof(sf::ChunkCodes, lens),
                         sizeof(sf::ChunkCodes), 320, bytes / 320, hipMemcpyDeviceToHost), "debug copy");
      return SFH_OK;
    }
    default: return SFH_E_INVALID_ARG;
  }
  if (bytes > avail) return SFH_E_INVALID_ARG;
  SF_HIP(hipSetDevice(ctx->device), "hipSetDevice");
  SF_HIP(hipMemcpy(host_dst, p, bytes, hipMemcpyDeviceToHost), "debug copy");
  return SFH_OK;
}

}  // extern "C"
