// microbenchmark: cost of unaligned ds_read_b32 / b64 / b128 versus aligned dwords + v_alignbyte
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_unaligned.hip -o /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t __attribute__((aligned(1))) u32u;
struct __attribute__((packed)) P16 { uint32_t a, b, c, d; };

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t* in, uint32_t* out, uint64_t* cyc, uint32_t mask) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s[];
  for (int i = threadIdx.x; i < 8192; i += 1024) ((uint32_t*)s)[i] = in[i];
  __syncthreads();
  uint32_t o = (in[threadIdx.x] * 2654435761u) >> 17;  // random byte offset < 32768
  o &= mask;
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
    if (MODE == 0) {  // 5 aligned dwords + 4 alignbyte (what cmp16 does)
      const uint32_t* d = (const uint32_t*)s + (o >> 2);
      const uint32_t c0 = d[0], c1 = d[1], c2 = d[2], c3 = d[3], c4 = d[4];
      acc += __builtin_amdgcn_alignbyte(c1, c0, o & 3) ^ __builtin_amdgcn_alignbyte(c2, c1, o & 3) ^
             __builtin_amdgcn_alignbyte(c3, c2, o & 3) ^ __builtin_amdgcn_alignbyte(c4, c3, o & 3);
    } else if (MODE == 1) {  // 4 unaligned dwords
      acc += *(const u32u*)(s + o) ^ *(const u32u*)(s + o + 4) ^ *(const u32u*)(s + o + 8) ^ *(const u32u*)(s + o + 12);
    } else {  // one unaligned 16-byte read
      const P16 q = *(const P16*)(s + o);
      acc += q.a ^ q.b ^ q.c ^ q.d;
    }
    o = (o + acc * 0 + 1237) & mask & 32767u;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  uint32_t *in, *out; uint64_t* cyc;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 512 * 8);
  uint32_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = i * 2654435761u + 12345;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[3] = {"5 aligned b32 + 4 alignbyte", "4 unaligned b32", "1 unaligned b128"};
  for (uint32_t mask : {0xFFFFFFFCu, 0xFFFFFFFFu}) {
    for (int m = 0; m < 3; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(1024), 36864, 0, in, out, cyc, mask);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(512), dim3(1024), 36864, 0, in, out, cyc, mask);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(512), dim3(1024), 36864, 0, in, out, cyc, mask);
      }
      hipDeviceSynchronize();
      uint64_t c[512]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
      uint64_t s = 0; for (int i = 0; i < 512; ++i) s += c[i];
      printf("offsets %s | %-30s : %.1f cycles per iteration per workgroup (16 waves)\n",
             mask == 0xFFFFFFFFu ? "byte-random " : "dword-random", names[m], (double)s / 512 / 256);
    }
  }
  return 0;
}
