// microbenchmark: sustained VALU issue rate (cycles per wave64 integer instruction per SIMD)
// at 1..8 waves per SIMD, for the integer ops the match step is made of.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int DEP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 1024; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (DEP) {  // one dependent chain
        a0 = __builtin_amdgcn_alignbyte(a0, a1, a0 & 3); a0 ^= a2; a0 = (a0 >> 3) + a3; a0 = a0 > a4 ? a0 : a5;
        a0 = __builtin_amdgcn_alignbyte(a0, a6, a0 & 3); a0 ^= a7; a0 = (a0 >> 3) + a1; a0 = a0 > a2 ? a0 : a3;
      } else {    // eight independent chains
        a0 = __builtin_amdgcn_alignbyte(a0, a1, seed); a1 ^= a2; a2 = (a2 >> 3) + a3; a3 = a3 > a4 ? a3 : a5;
        a4 = __builtin_amdgcn_alignbyte(a4, a5, seed); a5 ^= a6; a6 = (a6 >> 3) + a7; a7 = a7 > a0 ? a7 : a1;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* cyc;
  (void)hipMalloc(&out, 512 * 1024 * 4); (void)hipMalloc(&cyc, 512 * 8);
  for (int dep = 0; dep < 2; ++dep)
    for (int threads : {256, 512, 1024}) {
      for (int wgs = 1; wgs <= 2; ++wgs) {
        // wgs workgroups per CU: grid = 256 * wgs (one wave of workgroups)
        for (int rep = 0; rep < 2; ++rep) {
          if (dep) hipLaunchKernelGGL(k<1>, dim3(256 * wgs), dim3(threads), 0, 0, out, cyc, 3u);
          else hipLaunchKernelGGL(k<0>, dim3(256 * wgs), dim3(threads), 0, 0, out, cyc, 3u);
        }
        (void)hipDeviceSynchronize();
        uint64_t c[512]; (void)hipMemcpy(c, cyc, 256 * wgs * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * wgs; ++i) s += (double)c[i];
        s /= 256 * wgs;
        const double instr = 1024.0 * 4 * 10;  // ~10 VALU per unrolled body (8 + and/cmp)
        const double waves_per_simd = threads / 64.0 / 4.0 * wgs;
        printf("%s chains, %4d threads x %d WG/CU (%.0f waves/SIMD): %.2f cycles per wave-instr per wave, %.2f per SIMD\n",
               dep ? "dependent  " : "independent", threads, wgs, waves_per_simd, s / instr, s / instr / waves_per_simd);
      }
    }
  return 0;
}
