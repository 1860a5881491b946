// micro-test: in which order does the LDS execute the lanes of ONE ds_wrxchg_rtn_b32 wave-instruction when several lanes hit
// the same address?  If it is ascending lane order, `old = exchange(&T[h], lane)` hands every lane the nearest LOWER lane
// with the same address (or what the slot held before) -- an exact "previous occurrence" in one instruction, which is what
// a hash-chain insertion of 64 consecutive positions needs.
// What this file checks: ONE full-exec exchange per wave on a table private to the wave, over many random address patterns
// and collision densities, on every CU.  The check that mirrors the match kernels' real pattern -- partial exec masks,
// sixteen back-to-back instructions by one wave on buckets all slices share, the other waves reading (and, for
// ds_mskor_rtn_b32, storing into the other halves of) the same buckets meanwhile -- is the library's own
// starflate_amd/csrc/sf_guard.hip (sfh_lds_order_check; run per context before the first call that needs it, and by
// tests/test_gpu_parity.py at a few hundred million positions).  This file stays as the minimal stand-alone reproducer.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_xchg_order.hip -o tools/micro/lds_xchg_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k(uint32_t seed, uint32_t buckets, uint32_t iters, uint32_t* bad, uint32_t* shape) {
  __shared__ uint32_t T[4][1024];   // one table per wave
  __shared__ uint32_t H[4][64];     // the wave's addresses, for the reference model
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + threadIdx.x * 2246822519u;
  uint32_t nbad = 0, order_asc = 0, order_other = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    for (uint32_t i = lane; i < 1024; i += 64) T[wave][i] = 0xFFFF0000u | i;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t h = x % buckets;
    H[wave][lane] = h;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t old;
    asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"((uint32_t)((uint32_t)(uintptr_t)&T[wave][h])), "v"(lane) : "memory");
    // model: ascending lane order
    uint32_t want = 0xFFFF0000u | h;
    for (uint32_t l = 0; l < lane; ++l) if (H[wave][l] == h) want = l;
    if (old == want) ++order_asc; else { ++order_other; ++nbad; }
    // final content must be the LAST lane with that address
    __builtin_amdgcn_wave_barrier();
    uint32_t last = lane;
    for (uint32_t l = lane + 1; l < 64; ++l) if (H[wave][l] == h) last = l;
    if (T[wave][h] != last) ++nbad;
    __builtin_amdgcn_wave_barrier();
  }
  atomicAdd(&bad[0], nbad);
  atomicAdd(&shape[0], order_asc);
  atomicAdd(&shape[1], order_other);
}

int main() {
  uint32_t *bad, *shape;
  hipMalloc(&bad, 4); hipMalloc(&shape, 8);
  int rc = 0;
  for (uint32_t buckets : {1u, 2u, 3u, 7u, 16u, 33u, 64u, 200u, 1024u}) {
    hipMemset(bad, 0, 4); hipMemset(shape, 0, 8);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, 12345u + buckets, buckets, 200u, bad, shape);
    uint32_t hb = 0, hs[2] = {0, 0};
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hs, shape, 8, hipMemcpyDeviceToHost);
    printf("buckets %4u: lanes in ascending-order model %u, not %u, mismatches (incl. final content) %u\n", buckets, hs[0], hs[1], hb);
    if (hb) rc = 1;
  }
  printf(rc ? "NOT ascending lane order\n" : "ascending lane order holds\n");
  return rc;
}
