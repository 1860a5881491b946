// microbenchmark: cost of fetching the 8 bytes at a random LDS byte address c (a match candidate), per wave-instruction
// mix, 16 waves per workgroup, 2 workgroups per CU.  Modes:
//   0: three aligned ds_read_b32 + two v_alignbyte (round 2's rank8)
//   1: two 8-byte-aligned ds_read_b64 + selects + two v_alignbyte
//   2: one ds_read_b64 at a 4-byte-aligned address + one ds_read_b32 (inline asm; checks that the hardware returns
//      the right bytes for a b64 that is not 8-byte aligned)
//   3: ds_read2_b32 + ds_read_b32 (what the compiler makes of mode 2)
//   4: one 8-byte-aligned ds_read_b64 only (cost reference)   5: one ds_read_b32 only
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_gather.hip -o tools/micro/lds_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t ab(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(const uint32_t* in, uint32_t* out, uint64_t* cyc, uint32_t spread) {
  __shared__ __attribute__((aligned(16))) uint8_t s[40960];
  for (int i = threadIdx.x; i < 10240; i += 1024) ((uint32_t*)s)[i] = in[i] ;
  __syncthreads();
  const uint32_t* d32 = (const uint32_t*)s;
  uint32_t o = (in[threadIdx.x + 1024 * (blockIdx.x & 7)] * 2654435761u) >> 17;  // random byte offset < 32768
  if (spread == 0) o = 4096 + threadIdx.x;  // neighbours read neighbouring bytes (inside a long match)
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 2
  for (int it = 0; it < 512; ++it) {
    uint32_t a0, a1;
    if (MODE == 0) {
      const uint32_t w = o >> 2;
      const uint32_t c0 = d32[w], c1 = d32[w + 1], c2 = d32[w + 2];
      a0 = ab(c1, c0, o & 3); a1 = ab(c2, c1, o & 3);
    } else if (MODE == 1) {
      const uint2 p = *(const uint2*)(s + (o & ~7u)), q = *(const uint2*)(s + (o & ~7u) + 8);
      const bool up = (o & 4) != 0;
      const uint32_t c0 = up ? p.y : p.x, c1 = up ? q.x : p.y, c2 = up ? q.y : q.x;
      a0 = ab(c1, c0, o & 3); a1 = ab(c2, c1, o & 3);
    } else if (MODE == 2) {
      uint64_t p; uint32_t c2;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p), "=&v"(c2) : "v"(o & ~3u) : "memory");
      a0 = ab((uint32_t)(p >> 32), (uint32_t)p, o & 3); a1 = ab(c2, (uint32_t)(p >> 32), o & 3);
    } else if (MODE == 3) {
      uint64_t p; uint32_t c2;
      asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p), "=&v"(c2) : "v"(o & ~3u) : "memory");
      a0 = ab((uint32_t)(p >> 32), (uint32_t)p, o & 3); a1 = ab(c2, (uint32_t)(p >> 32), o & 3);
    } else if (MODE == 4) {
      const uint2 p = *(const uint2*)(s + (o & ~7u));
      a0 = p.x; a1 = p.y;
    } else {
      a0 = d32[o >> 2]; a1 = 0;
    }
    acc += a0 ^ (a1 * 3u);
    o = (o + ((acc & 0) | 1237u)) & 32767u;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  uint32_t *in, *out; uint64_t* cyc;
  const int G = 512;
  hipMalloc(&in, 10240 * 4); hipMalloc(&out, G * 1024 * 4); hipMalloc(&cyc, G * 8);
  static uint32_t h[10240]; for (int i = 0; i < 10240; ++i) h[i] = i * 2654435761u + 12345;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[6] = {"3 x b32 + 2 alignbyte", "2 x b64 (8B aligned) + 3 sel + 2 alignbyte", "b64 @4B-aligned + b32 (asm)",
                          "read2_b32 + b32 (asm)", "1 x b64 aligned only", "1 x b32 only"};
  static uint32_t res[6][G * 1024];
  for (uint32_t spread : {1u, 0u}) {
    for (int m = 0; m < 6; ++m) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) hipEventRecord(e0);
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
          default: hipLaunchKernelGGL(k<5>, dim3(G), dim3(1024), 0, 0, in, out, cyc, spread); break;
        }
      }
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      static uint64_t c[G]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
      hipMemcpy(res[m], out, sizeof res[m], hipMemcpyDeviceToHost);
      uint64_t sum = 0; for (int i = 0; i < G; ++i) sum += c[i];
      bool same = true;
      if (m >= 1 && m <= 3) for (int i = 0; i < G * 1024; ++i) if (res[m][i] != res[0][i]) { same = false; break; }
      // 2 workgroups per CU share it: cycles per wave-instruction-mix = wg cycles / (16 waves * 2 wgs) ... report raw too
      printf("%s | %-44s : %7.1f memtime ticks / iteration / workgroup, kernel %.3f ms%s\n", spread ? "random   " : "adjacent ",
             names[m], (double)sum / G / 512, ms, (m >= 1 && m <= 3) ? (same ? "  [same bytes as mode 0]" : "  [DIFFERENT bytes]") : "");
    }
  }
  return 0;
}
