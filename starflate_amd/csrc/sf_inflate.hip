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
//   k_inflate_bytes   the byte-copy half (src/decompress.cpp:157-187,388-398), one wave per segment with
//                     the 32 KiB window in LDS: 64 tokens per step, a wave prefix sum gives every token its
//                     output position; literals and short matches whose source lies before the step are
//                     written by their own lanes in parallel, the rest (long, or reading this step's own
//                     output) one after the other by the whole wave; the finished window leaves with
//                     16-byte stores.  A segment that is one stored block is copied straight from the stream.
//   k_inflate_status  first non-zero segment status in stream order = what the serial decoder would report.
#include "sf_device.h"
#include "sf_inflate_core.h"

namespace sf {

namespace {

constexpr uint32_t KT_LANES = 64;
constexpr uint32_t KT_LDS = KT_LANES * inflate::kLaneBytes;
constexpr uint32_t KB_SHORT = 16;  // matches up to this length are copied by their own lane

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

__global__ __launch_bounds__(64) void k_inflate_bytes(const uint8_t* __restrict__ src, uint64_t src_n,
                                                      const uint32_t* __restrict__ tokens, SegInfo* __restrict__ info,
                                                      uint8_t* __restrict__ dst) {
  __shared__ __align__(16) uint8_t win[kChunk + 64];
  const uint32_t seg = blockIdx.x, lane = threadIdx.x;
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
    for (uint32_t k = lane; k < nd; k += 64) {
      const uint32_t lo = load_word_guarded(src, src_n, w0 + k);
      const uint32_t hi = mis ? load_word_guarded(src, src_n, w0 + k + 1) : 0u;
      o32[k] = __builtin_amdgcn_alignbyte(hi, lo, mis);
    }
    const uint32_t done = 4 * nd;
    if (lane < out_n - done) o[done + lane] = src[si.raw_off + done + lane];
    return;
  }

  const uint32_t ntok = si.ntok;
  const uint32_t* tk = tokens + (uint64_t)seg * kChunk;
  uint32_t pos0 = 0;
  bool broken = false;
  uint32_t tok_next = lane < ntok ? tk[lane] : 0u;
  for (uint32_t g0 = 0; g0 < ntok; g0 += 64) {
    const uint32_t idx = g0 + lane;
    const bool valid = idx < ntok;
    const uint32_t tok = tok_next;
    tok_next = idx + 64 < ntok ? tk[idx + 64] : 0u;
    const bool is_m = valid && (tok >> 31);
    const uint32_t len = valid ? (is_m ? ((tok >> 16) & 0xFFu) + 3u : 1u) : 0u;
    const uint32_t incl = wave_scan_incl(len, lane);
    const uint32_t start = pos0 + incl - len;
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t dist = (tok & 0x7FFFu) + 1u;
    // k_inflate_tokens has validated every token; this only keeps a corrupted token buffer inside the window
    if (__any((is_m && dist > start) || (valid && start + len > out_n))) {
      broken = true;
      break;
    }
    if (valid && !is_m) win[start] = (uint8_t)tok;
    const uint32_t from = start - dist;
    const bool own = is_m && len <= KB_SHORT && from + len <= pos0;  // source entirely before this step
    {
      uint8_t b[KB_SHORT];
#pragma unroll
      for (uint32_t k = 0; k < KB_SHORT; ++k)
        if (own && k < len) b[k] = win[from + k];
#pragma unroll
      for (uint32_t k = 0; k < KB_SHORT; ++k)
        if (own && k < len) win[start + k] = b[k];
    }
    // the others in stream order, each by the whole wave: everything before such a match is final by then
    uint64_t rest = __ballot(is_m && !own);
    while (rest) {
      const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
      rest &= rest - 1;
      const uint32_t s = __builtin_amdgcn_readlane(start, l);
      const uint32_t f = __builtin_amdgcn_readlane(from, l);
      const uint32_t n = __builtin_amdgcn_readlane(len, l);
      const uint32_t d = s - f;
      if (d >= n || d >= 64) {
        // no overlap inside one 64-byte slice: slices in order (copy_from_before repeats, src/decompress.cpp:388-398)
        for (uint32_t k = lane; k < n; k += 64) win[s + k] = win[f + k];
      } else {
        for (uint32_t k = lane; k < n; k += 64) win[s + k] = win[f + k % d];  // period d, all sources final
      }
    }
    pos0 += total;
  }
  if (broken || pos0 != out_n) {
    if (lane == 0) info[seg].status = inflate::kError;
    return;
  }
  const uint4* w16 = reinterpret_cast<const uint4*>(win);
  uint4* o16 = reinterpret_cast<uint4*>(o);
  const uint32_t nq = out_n / 16;
  for (uint32_t k = lane; k < nq; k += 64) o16[k] = w16[k];
  const uint32_t done = 16 * nq;
  if (lane < out_n - done) o[done + lane] = win[done + lane];
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
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_tokens), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)KT_LDS);
}

hipError_t launch_inflate_tokens(const uint8_t* src, uint64_t src_n, const uint64_t* index, uint32_t nseg, uint64_t dst_n,
                                 uint32_t* tokens, SegInfo* info, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_tokens, dim3((nseg + KT_LANES - 1) / KT_LANES), dim3(KT_LANES), KT_LDS, s, src, src_n, index,
                     nseg, dst_n, tokens, info);
  return hipGetLastError();
}

hipError_t launch_inflate_bytes(const uint8_t* src, uint64_t src_n, uint32_t nseg, const uint32_t* tokens, SegInfo* info,
                                uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_bytes, dim3(nseg), dim3(64), 0, s, src, src_n, tokens, info, dst);
  return hipGetLastError();
}

hipError_t launch_inflate_status(const SegInfo* info, uint32_t nseg, uint32_t* d_result, hipStream_t s) {
  hipLaunchKernelGGL(k_inflate_status, dim3(1), dim3(KS_THREADS), 0, s, info, nseg, d_result);
  return hipGetLastError();
}

}  // namespace sf
