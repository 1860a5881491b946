// sf_checksum.hip -- CRC-32 (RFC 1952) and Adler-32 (RFC 1950) of the input on the GPU, and the
// zlib / gzip wrapper bytes around the raw DEFLATE stream (the reference's fixture
// tool deliberately strips the wrapper, /root/reference/tools/deflate_compress.py:8-13).
//
//   k_checksum<KIND>  one 256-thread workgroup per 32 KiB chunk, chunk staged through LDS
//                     (33-dword stride: conflict-free 128-byte thread segments); one u32 per chunk:
//                     CRC : remainder of chunk(x) * x^32 mod P with a zero register, the chunk
//                           zero-padded on the right to 32 KiB (so every chunk has the same length);
//                     Adler: (sum of weighted bytes mod 65521) << 16 | (sum of bytes mod 65521), weights
//                           counted from the padded chunk end.
//                     (CRC: table-driven per 128-byte thread segment, each remainder then multiplied by
//                     x^(8 * bytes behind it) mod P and all of them xor-ed)
//   k_wrap            one workgroup: folds the per-chunk values in order (CRC: a tree of multiplications
//                     by x^(8*length) mod P, then the right padding is divided out with x^-1 and the
//                     0xFFFFFFFF preset / final inversion are applied; Adler: plain modular sums),
//                     writes the wrapper header and trailer, bumps the stream size.
//
// All arithmetic is integer; results are bit-exact with zlib's crc32()/adler32().
#include "sf_device.h"

namespace sf {

namespace {

constexpr uint32_t KC_THREADS = 256;
constexpr uint32_t KC_SEG = kChunk / KC_THREADS;      // 128 bytes per thread
constexpr uint32_t KC_SEGW = KC_SEG / 4;              // 32 dwords
constexpr uint32_t KC_STAGE = KC_THREADS * (KC_SEGW + 1);
constexpr uint32_t KW_THREADS = 1024;
constexpr uint32_t kAdlerMod = 65521u;

// GF(2)[x] mod P, reflected representation: bit 31 is x^0 (as in the CRC register)
__host__ __device__ constexpr uint32_t gf2_mulx(uint32_t b) { return (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1; }
__host__ __device__ constexpr uint32_t gf2_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int k = 31; k >= 0; --k) {
    p ^= ((a >> k) & 1u) ? b : 0u;
    b = gf2_mulx(b);
  }
  return p;
}
__host__ __device__ constexpr uint32_t gf2_pow(uint32_t base, uint64_t e) {
  uint32_t r = 0x80000000u;
  for (; e; e >>= 1) {
    if (e & 1) r = gf2_mul(r, base);
    base = gf2_mul(base, base);
  }
  return r;
}
constexpr uint32_t kX8 = 0x00800000u;     // x^8
constexpr uint32_t kXinv = 0xDB710641u;   // x^-1: gf2_mulx(kXinv) == x^0
static_assert(gf2_mulx(kXinv) == 0x80000000u, "x^-1");

// shift[t] = x^(8 * 128 * (255 - t)): carries thread t's segment remainder to the end of the chunk
struct SegShift {
  uint32_t shift[KC_THREADS];
};
constexpr SegShift make_seg_shift() {
  SegShift s{};
  const uint32_t seg = gf2_pow(kX8, KC_SEG);
  uint32_t v = 0x80000000u;
  for (int t = KC_THREADS - 1; t >= 0; --t) {
    s.shift[t] = v;
    v = gf2_mul(v, seg);
  }
  return s;
}
__constant__ SegShift c_seg = make_seg_shift();
constexpr uint32_t kChunkOp = gf2_pow(kX8, kChunk);  // x^(8*32768): appends one chunk

template <uint32_t KIND>
__global__ __launch_bounds__(KC_THREADS) void k_checksum(const uint8_t* __restrict__ src, uint64_t n,
                                                         uint32_t* __restrict__ sums) {
  __shared__ uint32_t s_data[KC_STAGE];
  __shared__ uint32_t s_tab[KIND == kChecksumCrc32 ? 1024 : 1];
  __shared__ uint32_t s_part[KC_THREADS / 64][2];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint64_t cbase = (uint64_t)blockIdx.x * kChunk;
  const uint32_t valid = (uint32_t)(n - cbase < kChunk ? n - cbase : kChunk);  // n == 0: one empty chunk

  // stage: dword d of the chunk -> s_data[d + d/32]; bytes past the end of the input are zero
  const uint4* s16 = reinterpret_cast<const uint4*>(src + cbase);
#pragma unroll
  for (uint32_t i = 0; i < kChunk / 16 / KC_THREADS; ++i) {
    const uint32_t k = t + KC_THREADS * i, byte0 = 16 * k;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (byte0 + 16 <= valid) {
      q = s16[k];
    } else if (byte0 < valid) {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t b = 0; byte0 + b < valid; ++b) w[b >> 2] |= (uint32_t)src[cbase + byte0 + b] << (8 * (b & 3));
      q = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint32_t d = 4 * k, p = d + (d >> 5);
    s_data[p] = q.x;
    s_data[p + 1] = q.y;
    s_data[p + 2] = q.z;
    s_data[p + 3] = q.w;
  }

  if constexpr (KIND == kChecksumCrc32) {
    // slicing-by-4 tables: s_tab[256*j + b] = b(x) * x^(8*(j+1)) (register contents after j more zero bytes)
    uint32_t c = t;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = gf2_mulx(c);
    s_tab[t] = c;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 1; j < 4; ++j) {
      c = s_tab[c & 0xFF] ^ (c >> 8);
      s_tab[256 * j + t] = c;
    }
    __syncthreads();
    uint32_t r = 0;
    const uint32_t* seg = s_data + t * (KC_SEGW + 1);
#pragma unroll 8
    for (uint32_t j = 0; j < KC_SEGW; ++j) {
      r ^= seg[j];
      r = s_tab[768 + (r & 0xFF)] ^ s_tab[512 + ((r >> 8) & 0xFF)] ^ s_tab[256 + ((r >> 16) & 0xFF)] ^ s_tab[r >> 24];
    }
    // r(A||B) = r(A) * x^(8|B|) ^ r(B): every segment remainder is carried to the chunk end, then all are xor-ed
    r = gf2_mul(c_seg.shift[t], r);
#pragma unroll
    for (uint32_t o = 32; o; o >>= 1) r ^= __shfl_down(r, o);
    if (lane == 0) s_part[wave][0] = r;
    __syncthreads();
    if (t == 0) {
      uint32_t acc = 0;
      for (uint32_t w = 0; w < KC_THREADS / 64; ++w) acc ^= s_part[w][0];
      sums[blockIdx.x] = acc;
    }
  } else {
    __syncthreads();
    uint32_t a = 0, b = 0;
    const uint32_t* seg = s_data + t * (KC_SEGW + 1);
#pragma unroll 8
    for (uint32_t j = 0; j < KC_SEGW; ++j) {
      const uint32_t w = seg[j], wt = KC_SEG - 4 * j;  // weights wt, wt-1, wt-2, wt-3 for bytes 0..3
      const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF, b3 = w >> 24;
      a += b0 + b1 + b2 + b3;
      b += wt * b0 + (wt - 1) * b1 + (wt - 2) * b2 + (wt - 3) * b3;
    }
    // weights counted from the padded chunk end: segment t is followed by (255 - t) * 128 bytes
    uint32_t bb = (b + (kChunk - KC_SEG * (t + 1)) * a) % kAdlerMod;  // < 2.2e6 + 32640 * 32640 < 2^32
#pragma unroll
    for (uint32_t o = 32; o; o >>= 1) {
      a += __shfl_down(a, o);
      bb += __shfl_down(bb, o);
    }
    if (lane == 0) {
      s_part[wave][0] = a;
      s_part[wave][1] = bb;
    }
    __syncthreads();
    if (t == 0) {
      uint32_t sa = 0, sb = 0;
      for (uint32_t w = 0; w < KC_THREADS / 64; ++w) {
        sa += s_part[w][0];
        sb += s_part[w][1];
      }
      sums[blockIdx.x] = ((sb % kAdlerMod) << 16) | (sa % kAdlerMod);
    }
  }
}

// One workgroup.  total: in = header bytes + raw stream bytes (k_scan ran with that base), out += trailer.
// dst == nullptr: only *value is written (the checksum of the n input bytes).
__global__ __launch_bounds__(KW_THREADS) void k_wrap(const uint32_t* __restrict__ sums, uint32_t nchunks, uint64_t n,
                                                     uint32_t kind, uint8_t* __restrict__ dst,
                                                     uint64_t* __restrict__ total, uint32_t* __restrict__ value,
                                                     uint32_t chunk_op) {
  __shared__ uint32_t s_v[KW_THREADS];
  __shared__ uint32_t s_w[KW_THREADS];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (nchunks + KW_THREADS - 1) / KW_THREADS;
  const uint64_t c0 = (uint64_t)t * per;
  uint32_t result = 0;
  if (kind == kChecksumCrc32) {
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; ++k) {
      const uint64_t c = c0 + k;
      const uint32_t v = c < nchunks ? sums[c] : 0u;  // chunks past the end: zero padding
      acc = (acc ? gf2_mul(chunk_op, acc) : 0u) ^ v;
    }
    s_v[t] = acc;
    uint32_t op = gf2_pow(kX8, (uint64_t)kChunk * per);  // appends one thread's range
    __syncthreads();
    for (uint32_t s = 1; s < KW_THREADS; s <<= 1) {
      if ((t & (2 * s - 1)) == 0) s_v[t] = gf2_mul(op, s_v[t]) ^ s_v[t + s];
      op = gf2_mul(op, op);
      __syncthreads();
    }
    if (t == 0) {
      const uint64_t pad = (uint64_t)KW_THREADS * per * kChunk - n;  // zero bytes appended above
      const uint32_t raw = gf2_mul(s_v[0], gf2_pow(kXinv, 8 * pad));
      result = ~(raw ^ gf2_mul(gf2_pow(kX8, n), 0xFFFFFFFFu));  // preset register, final inversion
    }
  } else {
    uint64_t sa = 0, sb = 0;
    for (uint32_t k = 0; k < per; ++k) {
      const uint64_t c = c0 + k;
      if (c >= nchunks) break;
      const uint32_t v = sums[c];
      const uint64_t a = v & 0xFFFF, b = v >> 16;
      const int64_t d = (int64_t)n - (int64_t)((c + 1) * kChunk);  // bytes after the padded chunk end (< 0: padding)
      const uint64_t dm = (uint64_t)((d % (int64_t)kAdlerMod + (int64_t)kAdlerMod) % (int64_t)kAdlerMod);
      sa += a;
      sb = (sb + b + dm * a) % kAdlerMod;
    }
    s_v[t] = (uint32_t)(sa % kAdlerMod);
    s_w[t] = (uint32_t)sb;
    __syncthreads();
    for (uint32_t s = KW_THREADS / 2; s; s >>= 1) {
      if (t < s) {
        s_v[t] += s_v[t + s];  // <= 1024 * 65520 < 2^32
        s_w[t] += s_w[t + s];
      }
      __syncthreads();
    }
    if (t == 0) {
      const uint32_t a = (1u + s_v[0]) % kAdlerMod;
      const uint32_t b = (uint32_t)((n % kAdlerMod + s_w[0]) % kAdlerMod);
      result = (b << 16) | a;
    }
  }
  if (t != 0) return;
  if (value) *value = result;
  if (!dst) return;
  const uint64_t end = *total;
  uint8_t* tr = dst + end;
  if (kind == kChecksumAdler32) {
    dst[0] = 0x78;  // CM 8, CINFO 7 (32 KiB window)
    dst[1] = 0x9C;  // FLEVEL 2, no FDICT, FCHECK
    tr[0] = (uint8_t)(result >> 24);
    tr[1] = (uint8_t)(result >> 16);
    tr[2] = (uint8_t)(result >> 8);
    tr[3] = (uint8_t)result;
    *total = end + 4;
  } else {
    const uint8_t h[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF};  // no flags, MTIME 0, XFL 0, OS unknown
    for (int k = 0; k < 10; ++k) dst[k] = h[k];
    const uint32_t isize = (uint32_t)n;
    for (int k = 0; k < 4; ++k) {
      tr[k] = (uint8_t)(result >> (8 * k));
      tr[4 + k] = (uint8_t)(isize >> (8 * k));
    }
    *total = end + 8;
  }
}

}  // namespace

uint32_t wrapper_header_bytes(uint32_t kind) { return kind == kChecksumAdler32 ? 2u : kind == kChecksumCrc32 ? 10u : 0u; }

hipError_t launch_checksum(const uint8_t* src, uint64_t n, uint32_t nchunks, uint32_t kind, uint32_t* sums,
                           hipStream_t s) {
  if (kind == kChecksumCrc32)
    hipLaunchKernelGGL(k_checksum<kChecksumCrc32>, dim3(nchunks), dim3(KC_THREADS), 0, s, src, n, sums);
  else
    hipLaunchKernelGGL(k_checksum<kChecksumAdler32>, dim3(nchunks), dim3(KC_THREADS), 0, s, src, n, sums);
  return hipGetLastError();
}

hipError_t launch_wrap(const uint32_t* sums, uint32_t nchunks, uint64_t n, uint32_t kind, uint8_t* dst,
                       uint64_t* d_total, uint32_t* d_value, hipStream_t s) {
  hipLaunchKernelGGL(k_wrap, dim3(1), dim3(KW_THREADS), 0, s, sums, nchunks, n, kind, dst, d_total, d_value,
                     kChunkOp);
  return hipGetLastError();
}

uint32_t crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  return gf2_mul(gf2_pow(kX8, len_b), crc_a) ^ crc_b;
}

uint32_t adler32_combine(uint32_t adler_a, uint32_t adler_b, uint64_t len_b) {
  const uint64_t M = kAdlerMod;
  const uint64_t a1 = adler_a & 0xFFFF, b1 = adler_a >> 16, a2 = adler_b & 0xFFFF, b2 = adler_b >> 16;
  const uint64_t a = (a1 + a2 + M - 1) % M;
  const uint64_t b = (b1 + b2 + (len_b % M) * ((a1 + M - 1) % M)) % M;
  return (uint32_t)((b << 16) | a);
}

}  // namespace sf
