// sf_guard.hip -- run-time check of the one piece of LDS behaviour the exact-recency match finders rest on and the ISA
// manual does not promise: a returning LDS atomic (ds_wrxchg_rtn_b32: the chain efforts; ds_mskor_rtn_b32: SFH_EFFORT_RECENT)
// executes the lanes of ONE wave-instruction in ASCENDING LANE ORDER where they meet at one address, and one wave's
// instructions in the order they were issued.  Then `old = exchange(&bucket[hash], own code)`, issued slice after slice by
// one wave, hands every position the nearest earlier position with its hash -- an exact hash-chain insertion of 64
// positions per instruction (k_lz77, sf_kernels.hip).  A device that orders differently would still produce valid streams
// (candidates are byte-compared) but not the specification's, so the library checks before the first call that needs it
// (sfh_lds_order_check; cached per context) and refuses those efforts where the check fails.
//
// The kernel mirrors how k_lz77 uses the instruction, not a simplification of it: 1024 threads post a hash each (some
// lanes sit a step out: partial exec masks), ONE wave issues the sixteen slices' atomics back to back on a table all
// sixteen slices share, the fifteen other waves read the same table and the posts meanwhile, and every position then
// compares what it was handed -- and what the table holds at the end -- with the sequential model.  With op 1 the other
// waves also do what the posting half of k_lz77 does beside the pass: 16-bit stores into the HIGH halves of the very
// buckets whose low halves the masked atomics replace; neither may lose the other's update.
#include "sf_device.h"

namespace sf {

constexpr uint32_t G_THREADS = 1024, G_BUCKETS = 2048;
// initial bucket contents: told apart from every code (1..1024) and from each other
__device__ __forceinline__ uint32_t guard_init(uint32_t i) { return ((0x8000u | i) << 16) | (0x4000u | i); }

template <int OP>  // 0: ds_wrxchg_rtn_b32 (whole dword), 1: ds_mskor_rtn_b32 with mask 0xFFFF (the low half only)
__global__ __launch_bounds__(G_THREADS) void k_lds_order(uint32_t seed, uint32_t buckets, uint32_t iters,
                                                         uint32_t* __restrict__ result /* [0] mismatches, [1] positions checked */) {
  __shared__ uint32_t T[G_BUCKETS];
  __shared__ uint16_t post[G_THREADS];
  __shared__ uint32_t ans[G_THREADS];
  __shared__ uint32_t sink[16];
  const uint32_t t = threadIdx.x, lane = t & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + t * 2246822519u + 1u;
  uint32_t nbad = 0, nchecked = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    for (uint32_t i = t; i < G_BUCKETS; i += G_THREADS) T[i] = guard_init(i);
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t h = x % buckets;
    const bool ins = ((x >> 20) & 7u) != 0;  // one lane in eight sits the step out
    post[t] = (uint16_t)(h | (ins ? 0x8000u : 0u));
    __syncthreads();
    if (wave == (it & 15u)) {
      uint32_t e[16], old[16];
#pragma unroll
      for (uint32_t sl = 0; sl < 16; ++sl) e[sl] = post[sl * 64 + lane];
#pragma unroll
      for (uint32_t sl = 0; sl < 16; ++sl) {
        old[sl] = 0xFFFFFFFFu;
        const uint32_t addr = (uint32_t)(uintptr_t)&T[e[sl] & (G_BUCKETS - 1)];
        const uint32_t code = sl * 64 + lane + 1;
        if (e[sl] & 0x8000u) {
          // ("+v": the register that holds the lane's default IS the one the LDS writes to, so nothing copies it early)
          if (OP == 0) asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "+v"(old[sl]) : "v"(addr), "v"(code) : "memory");
          else asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "+v"(old[sl]) : "v"(addr), "v"(0xFFFFu), "v"(code) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (uint32_t sl = 0; sl < 16; ++sl) {
        asm volatile("" : "+v"(old[sl]));
        ans[sl * 64 + lane] = old[sl];
      }
    } else {
      // the other waves keep the LDS busy with reads of the same table and of the posts
      uint32_t acc = 0, y = x ^ (it * 747796405u);
#pragma unroll 4
      for (uint32_t k = 0; k < 24; ++k) {
        y ^= y << 13; y ^= y >> 17; y ^= y << 5;
        acc += T[(y >> 7) & (G_BUCKETS - 1)] + post[y & (G_THREADS - 1)];
      }
      if (acc == 0x12345u) sink[wave] = acc;
      // op 1: the high half of the thread's own bucket, while the pass is under way (some before it, some behind)
      if (OP == 1 && ins) *reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(&T[h]) + 2) = (uint16_t)(guard_init(h) & 0xFFFFu);
    }
    __syncthreads();
    const uint32_t got = ans[t];
    if (!ins) {
      nbad += got != 0xFFFFFFFFu;
    } else {
      // sequential model: positions go in one by one in ascending order
      const uint32_t key = h | 0x8000u;
      uint32_t prev = 0, later = 0;
      for (uint32_t p = t; p-- > 0;)
        if (post[p] == key) { prev = p + 1; break; }
      for (uint32_t p = t + 1; p < G_THREADS; ++p)
        if (post[p] == key) { later = 1; break; }
      const uint32_t init = guard_init(h);
      uint32_t want, fin;
      if (OP == 0) {
        want = prev ? prev : init;
        fin = t + 1;
        nbad += got != want;
      } else {
        // the returned high half is the old or the new one, whichever the store beside the pass had made of it by then
        want = prev ? prev : (init & 0xFFFFu);
        const uint32_t ghi = got >> 16;
        nbad += (got & 0xFFFFu) != want || (ghi != (init >> 16) && ghi != (init & 0xFFFFu));
        // at the end: high half = what the stores put there (the serial wave's own positions store nothing: the bucket
        // then keeps its old high half unless a position of another wave shares it), low half = the last position
        fin = t + 1;
      }
      if (!later) {
        const uint32_t cur = T[h];
        if (OP == 0) nbad += cur != fin;
        else {
          uint32_t stored = 0;  // did any position of a storing wave post this bucket?
          for (uint32_t p = 0; p < G_THREADS; ++p)
            if (post[p] == key && (p >> 6) != (it & 15u)) { stored = 1; break; }
          nbad += (cur & 0xFFFFu) != fin || (cur >> 16) != (stored ? (init & 0xFFFFu) : (init >> 16));
        }
      }
      ++nchecked;
    }
    __syncthreads();
  }
  if (nbad) atomicAdd(&result[0], nbad);
  atomicAdd(&result[1], nchecked);
}

// Runs the check: `blocks` workgroups x `iters` steps x five collision densities.  *mismatches / *checked: totals.
hipError_t run_lds_order_check(int op, uint32_t blocks, uint32_t iters, uint32_t* d_result, hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_result, 0, 2 * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  const uint32_t dens[5] = {1u, 5u, 64u, 700u, G_BUCKETS};
  for (uint32_t k = 0; k < 5; ++k) {
    if (op == 0) hipLaunchKernelGGL(k_lds_order<0>, dim3(blocks), dim3(G_THREADS), 0, s, 977u + 31u * k, dens[k], iters, d_result);
    else hipLaunchKernelGGL(k_lds_order<1>, dim3(blocks), dim3(G_THREADS), 0, s, 977u + 31u * k, dens[k], iters, d_result);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace sf
