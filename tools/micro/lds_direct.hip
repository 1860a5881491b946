// microbenchmark / semantics probe: global_load_lds_dwordx4 on gfx950 -- where does lane L's data land in LDS?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_direct.hip -o /tmp/lds_direct
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* __restrict__ g, uint32_t* out, int masked) {
  __shared__ __attribute__((aligned(16))) uint32_t s[2][64 * 4];
  for (int i = threadIdx.x; i < 2 * 64 * 4; i += 64) (&s[0][0])[i] = 0xDEAD0000u + i;
  __syncthreads();
  const uint32_t* p = g + threadIdx.x * 7;  // dword-aligned, not 16-byte aligned
  if (!masked || (threadIdx.x & 1))
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)&s[1][0], 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 64 * 4; i += 64) out[i] = (&s[0][0])[i];
}
int main() {
  uint32_t *g, *o;
  (void)hipMalloc(&g, 4096 * 4); (void)hipMalloc(&o, 512 * 4);
  uint32_t h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
  (void)hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  for (int masked = 0; masked < 2; ++masked) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, masked);
    uint32_t r[512]; (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int ok = 1, untouched0 = 1;
    for (int i = 0; i < 256; ++i) untouched0 &= r[i] == 0xDEAD0000u + i;
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 4; ++c) {
        const uint32_t want = (!masked || (l & 1)) ? (uint32_t)(l * 7 + c) : 0xDEAD0000u + 256 + l * 4 + c;
        if (r[256 + l * 4 + c] != want) { if (ok) printf("masked=%d first mismatch lane %d c %d: got %08x want %08x\n", masked, l, c, r[256 + l * 4 + c], want); ok = 0; }
      }
    printf("masked=%d: lane L's 16 bytes at base + 16*L: %s; buffer 0 untouched: %s\n", masked, ok ? "yes" : "NO", untouched0 ? "yes" : "NO");
  }
  return 0;
}
