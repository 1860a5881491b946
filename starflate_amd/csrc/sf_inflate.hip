// sf_inflate.hip -- GPU decode of block-indexed DEFLATE streams: the reference's
// decompress() (/root/reference/src/decompress.cpp:402-461) for streams whose independently decodable
// segments are known -- every stream this library writes (one segment per 32 KiB chunk, byte-aligned, no
// match reaching before the segment; the index is the chunk offset table k_scan already produces), and
// e.g. zlib streams flushed with Z_FULL_FLUSH every 32 KiB.  Arbitrary DEFLATE is serial (README.md:5-6);
// the index is what makes it parallel.
//
//   k_inflate_tokens  the bit-serial half.  Huffman decoding cannot be split inside a segment, so the SIMT
//                     mapping is one LANE per segment: 64 segments per wave, each lane running
//                     decode_segment() (sf_inflate_core.h) with its own code tables in a 2,148-byte slice
//                     of LDS (137 KiB per workgroup) and its own 64-bit bit buffer fed by dword loads one
//                     refill ahead.  Output: the k_lz77 token format, four tokens per 16-byte store.
//   k_inflate_bytes   the byte-copy half (src/decompress.cpp:157-187,388-398), one 1024-thread workgroup per
//                     segment, 96 KiB of LDS: a workgroup prefix sum over the token lengths places every
//                     token; each output byte gets a 16-bit pointer (a literal to itself, a match byte to the
//                     byte `distance` before it); pointer jumping resolves all copy chains at once in <= 15
//                     barrier-separated rounds, whatever the nesting or overlap; every byte then fetches its
//                     literal and the window leaves with 16-byte stores.  A segment that is one stored block
//                     is copied straight from the stream.
//   k_inflate_status  first non-zero segment status in stream order = what the serial decoder would report.
#include "sf_device.h"
#include "sf_inflate_core.h"

namespace sf {

namespace {

constexpr uint32_t KT_LANES = 64;
constexpr uint32_t KT_LDS = KT_LANES * inflate::kLaneBytes;
constexpr uint32_t KB_THREADS = 1024;
constexpr uint32_t KB_LDS = 3 * kChunk + 128;  // pointers (u16) + bytes + scan scratch
constexpr uint32_t KB_SHORT = 16;  // a thread writes this many pointers of its match itself, the wave the rest

__global__ __launch_bounds__(KT_LANES) void k_inflate_tokens(const uint8_t* __restrict__ src, uint64_t src_n,
                                                            const uint64_t* __restrict__ index, uint32_t nseg,
                                                            uint64_t dst_n, uint32_t* __restrict__ tokens,
                                                            SegInfo* __restrict__ info) {
  extern __shared__ __align__(16) uint8_t s_tables[];
  const uint32_t seg = blockIdx.x * KT_LANES + threadIdx.x;
  if (seg >= nseg) return;
  const uint64_t lo = index[seg], hi = index[seg + 1];
  const uint64_t obase = (uint64_t)seg * kChunk;
  const uint32_t out_n = dst_n > obase ? (uint32_t)(dst_n - obase < kChunk ? dst_n - obase : kChunk) : 0u;
  const inflate::SegmentResult r = inflate::decode_segment(src, src_n, lo, hi, out_n, tokens + (uint64_t)seg * kChunk,
                                                           s_tables + threadIdx.x * inflate::kLaneBytes);
  SegInfo si;
  si.status = r.status;
  si.ntok = r.ntok;
  si.raw = r.raw;
  si.out_n = out_n;
  si.raw_off = r.raw_off;
  info[seg] = si;
}

// Streams of this library: one block per segment and a sub-index naming, for each of the 32 parse regions
// (1024 bytes of output; k_lz77 never lets a match cross them), the bit offset of the region's first token
// code and the number of tokens before it (k_emit writes both).  One wave per segment: lane 0 reads the block
// header and builds the code tables once, in LDS; lanes 0..31 then decode one region each, 32 bit streams of
// the same block side by side, writing tokens at their compact positions.  The sub-index is checked against the
// stream (first code right after the header, every lane ends exactly where the next begins, exact byte and
// token counts): a wrong sub-index is an error, never wrong output.
__global__ __launch_bounds__(64) void k_inflate_tokens_sub(const uint8_t* __restrict__ src, uint64_t src_n,
                                                          const uint64_t* __restrict__ index,
                                                          const uint32_t* __restrict__ subidx, uint64_t dst_n,
                                                          uint32_t* __restrict__ tokens, SegInfo* __restrict__ info) {
  __shared__ __align__(16) uint8_t s_tab[inflate::kLaneBytes + 12];
  __shared__ uint32_t s_open[2];
  __shared__ uint64_t s_open64[2];
  const uint32_t seg = blockIdx.x, lane = threadIdx.x;
  const uint64_t lo = index[seg], hi = index[seg + 1];
  const uint64_t obase = (uint64_t)seg * kChunk;
  const uint32_t out_n = dst_n > obase ? (uint32_t)(dst_n - obase < kChunk ? dst_n - obase : kChunk) : 0u;
  if (lane == 0) {
    uint32_t raw;
    uint64_t raw_off, hdr_end;
    s_open[0] = inflate::open_segment(src, src_n, lo, hi, out_n, s_tab, raw, raw_off, hdr_end);
    s_open[1] = raw;
    s_open64[0] = raw_off;
    s_open64[1] = hdr_end;
  }
  __syncthreads();
  uint32_t status = s_open[0];
  const uint32_t raw = s_open[1];
  uint32_t ntok = 0;
  if (status == inflate::kOk && !raw) {
    const uint32_t* sub = subidx + (uint64_t)seg * 2 * kSubRegions;
    uint32_t st = inflate::kOk, n = 0, tok0 = 0;
    if (lane < kSubRegions) {
      const uint32_t bit0 = sub[2 * lane];
      tok0 = sub[2 * lane + 1];
      const uint32_t bit1 = lane + 1 < kSubRegions ? sub[2 * lane + 2] : 0u;
      const uint32_t tok1 = lane + 1 < kSubRegions ? sub[2 * lane + 3] : 0u;
      const uint32_t ob = lane * kRegion < out_n ? lane * kRegion : out_n;
      const uint32_t oe = (lane + 1) * kRegion < out_n ? (lane + 1) * kRegion : out_n;
      if ((lane == 0 && bit0 != s_open64[1]) || tok0 > ob) {
        st = inflate::kError;  // (tokens before a region) <= (bytes before it) also bounds the token stores
      } else {
        st = inflate::decode_region(src, src_n, lo, hi, bit0, bit1, lane + 1 == kSubRegions, ob, oe,
                                    tokens + (uint64_t)seg * kChunk + tok0, s_tab, n);
        if (st == inflate::kOk && lane + 1 < kSubRegions && tok0 + n != tok1) st = inflate::kError;
      }
    }
    const uint64_t failed = __ballot(st != inflate::kOk);
    if (failed) status = __builtin_amdgcn_readlane(st, __builtin_amdgcn_readfirstlane(__builtin_ctzll(failed)));
    ntok = __builtin_amdgcn_readlane(tok0 + n, kSubRegions - 1);
  }
  if (lane == 0) {
    SegInfo si;
    si.status = status;
    si.ntok = ntok;
    si.raw = raw;
    si.out_n = out_n;
    si.raw_off = s_open64[0];
    info[seg] = si;
  }
}

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t lane) {
#pragma unroll
  for (uint32_t o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  return v;
}

__device__ __forceinline__ uint32_t load_word_guarded(const uint8_t* base, uint64_t src_n, uint64_t w) {
  const uint64_t b = 4 * w;
  if (b + 4 <= src_n) return reinterpret_cast<const uint32_t*>(base)[w];
  uint32_t v = 0;
  for (uint32_t k = 0; k < 4; ++k)
    if (b + k < src_n) v |= (uint32_t)base[b + k] << (8 * k);
  return v;
}

// The byte-copy half.  A match copies bytes that may themselves come from a match, so a serial decoder
// (src/decompress.cpp:157-187,388-398) is a chain of dependent copies.  Here every output byte gets a
// pointer instead -- a literal points to itself, byte k of a match to the byte `distance` before it -- and
// pointer jumping (ptr[j] = ptr[ptr[j]]) resolves all chains of a segment at once in at most log2(32768)
// barrier-separated rounds, however the matches nest or overlap; then every byte reads its literal.
__global__ __launch_bounds__(KB_THREADS) void k_inflate_bytes(const uint8_t* __restrict__ src, uint64_t src_n,
                                                              const uint32_t* __restrict__ tokens,
                                                              SegInfo* __restrict__ info, uint8_t* __restrict__ dst) {
  extern __shared__ __align__(16) uint8_t s_dyn[];
  uint16_t* s_ptr = reinterpret_cast<uint16_t*>(s_dyn);          // [32768]
  uint8_t* s_byte = s_dyn + 2 * kChunk;                          // [32768] literals, then the output
  uint32_t* s_wtot = reinterpret_cast<uint32_t*>(s_dyn + 3 * kChunk);  // [16] wave totals, [16] flag
  const uint32_t seg = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const SegInfo si = info[seg];
  if (si.status != inflate::kOk) return;
  const uint32_t out_n = si.out_n;
  uint8_t* o = dst + (uint64_t)seg * kChunk;  // 16-byte aligned

  if (si.raw) {
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
    return;
  }

  // ---- tokens -> pointers: 1024 tokens per step, a workgroup prefix sum places them ----
  const uint32_t ntok = si.ntok;
  const uint32_t* tk = tokens + (uint64_t)seg * kChunk;
  uint32_t pos0 = 0;
  bool bad = false;
  for (uint32_t g0 = 0; g0 < ntok; g0 += KB_THREADS) {
    const uint32_t idx = g0 + t;
    const bool valid = idx < ntok;
    const uint32_t tok = valid ? tk[idx] : 0u;
    const bool is_m = valid && (tok >> 31);
    const uint32_t len = valid ? (is_m ? ((tok >> 16) & 0xFFu) + 3u : 1u) : 0u;
    const uint32_t incl = wave_scan_incl(len, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < KB_THREADS / 64; ++w) {
      const uint32_t v = s_wtot[w];
      if (w < wave) pre += v;
      all += v;
    }
    const uint32_t start = pos0 + pre + incl - len;
    const uint32_t dist = (tok & 0x7FFFu) + 1u;
    // k_inflate_tokens has validated every token; this keeps a corrupted token buffer inside the window
    if ((is_m && dist > start) || (valid && start + len > out_n)) bad = true;
    else if (valid) {
      if (!is_m) {
        s_ptr[start] = (uint16_t)start;
        s_byte[start] = (uint8_t)tok;
      } else {
        const uint32_t from = start - dist;
        const uint32_t n0 = len < KB_SHORT ? len : KB_SHORT;
        // a source byte placed by an earlier step already carries a resolved (or at least shortened) pointer:
        // adopt it, so that the jumping below only has to untangle chains inside one step
        for (uint32_t k = 0; k < n0; ++k) {
          const uint32_t sp = from + k;
          s_ptr[start + k] = sp < pos0 ? s_ptr[sp] : (uint16_t)sp;
        }
      }
    }
    // the long tails, one match after the other by the whole wave
    uint64_t rest = __ballot(is_m && !bad && len > KB_SHORT);
    while (rest) {
      const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
      rest &= rest - 1;
      const uint32_t s0 = __builtin_amdgcn_readlane(start, l);
      const uint32_t d0 = __builtin_amdgcn_readlane(dist, l);
      const uint32_t n = __builtin_amdgcn_readlane(len, l);
      for (uint32_t k = KB_SHORT + lane; k < n; k += 64) {
        const uint32_t sp = s0 + k - d0;
        s_ptr[s0 + k] = sp < pos0 ? s_ptr[sp] : (uint16_t)sp;
      }
    }
    pos0 += all;
    __syncthreads();  // s_wtot is rewritten by the next step
  }
  if (__syncthreads_or(bad || pos0 != out_n)) {
    if (t == 0) info[seg].status = inflate::kError;
    return;
  }

  // ---- pointer jumping, in place: a pointer only ever moves to an ancestor, so mixed old/new reads are fine ----
  for (uint32_t round = 0; round < 16; ++round) {
    bool changed = false;
#pragma unroll 4
    for (uint32_t j = t; j < out_n; j += KB_THREADS) {
      const uint32_t p = s_ptr[j];
      if (p != j) {
        const uint32_t q = s_ptr[p];
        if (q != p) {
          s_ptr[j] = (uint16_t)q;
          changed = true;
        }
      }
    }
    if (!__syncthreads_or(changed)) break;
  }
  // every pointer now names a literal position; literal positions keep their own byte, so in place is safe
  for (uint32_t j = t; j < out_n; j += KB_THREADS) {
    const uint32_t p = s_ptr[j];
    if (p != j) s_byte[j] = s_byte[p];
  }
  __syncthreads();
  const uint4* w16 = reinterpret_cast<const uint4*>(s_byte);
  uint4* o16 = reinterpret_cast<uint4*>(o);
  const uint32_t nq = out_n / 16;
  for (uint32_t k = t; k < nq; k += KB_THREADS) o16[k] = w16[k];
  const uint32_t done = 16 * nq;
  if (t < out_n - done) o[done + t] = s_byte[done + t];
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

hipError_t launch_inflate_tokens(const uint8_t* src, uint64_t src_n, const uint64_t* index, uint32_t nseg, uint64_t dst_n,
                                 uint32_t* tokens, SegInfo* info, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_tokens, dim3((nseg + KT_LANES - 1) / KT_LANES), dim3(KT_LANES), KT_LDS, s, src, src_n, index,
                     nseg, dst_n, tokens, info);
  return hipGetLastError();
}

hipError_t launch_inflate_tokens_sub(const uint8_t* src, uint64_t src_n, const uint64_t* index, const uint32_t* subidx,
                                     uint32_t nseg, uint64_t dst_n, uint32_t* tokens, SegInfo* info, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_tokens_sub, dim3(nseg), dim3(64), 0, s, src, src_n, index, subidx, dst_n, tokens, info);
  return hipGetLastError();
}

hipError_t launch_inflate_bytes(const uint8_t* src, uint64_t src_n, uint32_t nseg, const uint32_t* tokens, SegInfo* info,
                                uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_bytes, dim3(nseg), dim3(KB_THREADS), KB_LDS, s, src, src_n, tokens, info, dst);
  return hipGetLastError();
}

hipError_t launch_inflate_status(const SegInfo* info, uint32_t nseg, uint32_t* d_result, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_status, dim3(1), dim3(KS_THREADS), 0, s, info, nseg, d_result);
  return hipGetLastError();
}

}  // namespace sf
