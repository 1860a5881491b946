// microbenchmark: issue cost of the integer VALU instructions k_lz77 is made of, in core cycles per wave64 instruction
// per SIMD with 8 waves per SIMD (2 workgroups x 1024 threads per CU, as k_lz77 runs), independent chains.
// Timed with HIP events over the whole launch; cycles = time x the device's current clock (hipDeviceAttributeClockRate
// is the maximum; the measured s_memtime rate is printed beside it).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_ops.hip -o tools/micro/valu_ops
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(1024, 8) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;
  const uint32_t s = seed & 3;
  const uint64_t mask = 0x5555555555555555ull ^ seed;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    // 8 independent chains x 8 = 64 instructions of the kind per iteration
#define ONE(r, x, y)                                                                                                   \
  if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(x));                                                \
  if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(x));                                                \
  if (OP == 2) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                              \
  if (OP == 3) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r));                                                            \
  if (OP == 4) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                   \
  if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x));                                       \
  if (OP == 6) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                \
  if (OP == 7) asm volatile("v_bfe_u32 %0, %0, %1, 10" : "+v"(r) : "v"(x));                                            \
  if (OP == 8) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(r) : "v"(x));                                         \
  if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(x));                                             \
  if (OP == 10) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(r), "v"(x) : "vcc");                                     \
  if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(x));         \
  if (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(x));                                           \
  if (OP == 13) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r) : "v"(x));                                          \
  if (OP == 14) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(*(uint64_t*)&r##r) : "v"(x));                           \
  if (OP == 15) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(x));                                            \
  if (OP == 16) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                \
  if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                  \
  if (OP == 18) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                               \
  if (OP == 19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                  \
  if (OP == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(x));                                               \
  if (OP == 21) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(x));                                                \
  if (OP == 22) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(x));                                               \
  if (OP == 23) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r) : "v"(y));                                           \
  if (OP == 24) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r) : "v"(y));                                           \
  if (OP == 25) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(x));                                               \
  if (OP == 26) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(x));                                                   \
  if (OP == 27) asm volatile("v_not_b32 %0, %0" : "+v"(r));                                                            \
  if (OP == 28) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x) : "vcc");  \
  if (OP == 29) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(x), "s"(mask));                        \
  if (OP == 30) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(x), "v"(y));                     \
  if (OP == 31) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                              \
  if (OP == 32) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                   \
  if (OP == 33) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                    \
  if (OP == 34) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(x));                                       \
  if (OP == 35) asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(x));      \
  if (OP == 36) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r) : "v"(x)); \
  if (OP == 37) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                    \
  if (OP == 38) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(x));                                               \
  if (OP == 39) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));                                    \
  if (OP == 40) asm volatile("v_and_b32_e64 %0, %0, %1" : "+v"(r) : "s"((uint32_t)mask));                                         \
  if (OP == 41) asm volatile("v_cmp_lt_u32_e64 %1, %0, %2" : "+v"(r), "=s"(sm) : "v"(x));                               \
  if (OP == 42) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(r) : "v"(x));                                        \
  if (OP == 43) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(*(uint64_t*)&r##r) : "v"(*(uint64_t*)&r##r));         \
  if (OP == 44) asm volatile("v_readlane_b32 %1, %0, 3\n\ts_nop 3" : "+v"(r), "=s"(sl));                                \
  if (OP == 45) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(r) : "s"((uint32_t)mask));                                           \
  if (OP == 46) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));               \
  if (OP == 47) asm volatile("v_mov_b32 %0, 0x12345678" : "=v"(r));                                                     \
  if (OP == 48) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(*(uint64_t*)&r##r) : "v"(x));
    uint64_t aa = a, bb = b, cc = c, dd = d, ee = e, ff = f, gg = g, hh = h, sm = 0;
    uint32_t sl = 0;
    (void)sm; (void)sl;
    (void)aa; (void)bb; (void)cc; (void)dd; (void)ee; (void)ff; (void)gg; (void)hh;
    REP8(ONE(a, b, s) ONE(b, c, s) ONE(c, d, s) ONE(d, e, s) ONE(e, f, s) ONE(f, g, s) ONE(g, h, s) ONE(h, a, s))
    if (OP == 14 || OP == 43 || OP == 48) { a ^= (uint32_t)aa; b ^= (uint32_t)bb; c ^= (uint32_t)cc; d ^= (uint32_t)dd; e ^= (uint32_t)ee; f ^= (uint32_t)ff; g ^= (uint32_t)gg; h ^= (uint32_t)hh; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

__global__ void k_clock(uint64_t* o) {
  const uint64_t m0 = __builtin_amdgcn_s_memtime(), c0 = __builtin_readcyclecounter();
  for (int i = 0; i < 200000; ++i) asm volatile("s_nop 0");
  const uint64_t m1 = __builtin_amdgcn_s_memtime(), c1 = __builtin_readcyclecounter();
  o[0] = m1 - m0; o[1] = c1 - c0;
}

template <int OP>
void run(const char* name, uint32_t* out, double ghz) {
  const int iters = 2000, grid = 512;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(1024), 0, 0, out, 3u, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(1024), 0, 0, out, 3u, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 8 waves x iters x 64 instructions
  const double inst = 8.0 * iters * 64;
  printf("%-28s %.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at %.2f GHz\n", name, ms, ms * 1e6 / inst,
         ms * 1e6 / inst * ghz, ghz);
}

int main() {
  uint32_t* out; uint64_t* o;
  (void)hipMalloc(&out, 512 * 1024 * 4); (void)hipMalloc(&o, 16);
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, o);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, o);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[2]; (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("clock attribute %.3f GHz; in one idle-ish kernel of %.3f ms: s_memtime %.1f MHz, cycle counter (s_memrealtime/clock64) %.1f MHz\n",
         khz / 1e6, ms, h[0] / (ms * 1e3), h[1] / (ms * 1e3));
  const double ghz = khz / 1e6;
  run<0>("v_add_u32 (VOP2)", out, ghz); run<1>("v_xor_b32 (VOP2)", out, ghz); run<2>("v_alignbyte_b32 (VOP3)", out, ghz);
  run<3>("v_ffbl_b32 (VOP1)", out, ghz); run<4>("v_min3_u32 (VOP3)", out, ghz); run<5>("v_cndmask_b32 (VOP2, vcc)", out, ghz);
  run<6>("v_mad_i32_i24 (VOP3)", out, ghz); run<7>("v_bfe_u32 (VOP3)", out, ghz); run<8>("v_lshl_or_b32 (VOP3)", out, ghz);
  run<9>("v_mul_lo_u32", out, ghz); run<10>("v_cmp_lt_u32 (vcc)", out, ghz); run<11>("v_mov_b32_dpp wave_shr:1", out, ghz);
  run<12>("v_mul_u32_u24", out, ghz); run<13>("v_bcnt_u32_b32", out, ghz); run<14>("v_lshlrev_b64", out, ghz);
  run<15>("v_pk_add_u16", out, ghz); run<16>("v_and_or_b32", out, ghz); run<17>("v_perm_b32", out, ghz);
  run<18>("v_dot4_u32_u8", out, ghz); run<19>("v_add3_u32", out, ghz);
  run<20>("v_and_b32 (VOP2)", out, ghz); run<21>("v_or_b32 (VOP2)", out, ghz); run<22>("v_sub_u32 (VOP2)", out, ghz);
  run<23>("v_lshlrev_b32 (VOP2)", out, ghz); run<24>("v_lshrrev_b32 (VOP2)", out, ghz); run<25>("v_min_u32 (VOP2)", out, ghz);
  run<26>("v_mov_b32 (VOP1)", out, ghz); run<27>("v_not_b32 (VOP1)", out, ghz); run<28>("v_cmp + v_cndmask (PAIR: 2 instr)", out, ghz);
  run<29>("v_cndmask_b32_e64 (sgpr mask)", out, ghz); run<30>("v_bitop3_b32", out, ghz); run<31>("v_alignbit_b32", out, ghz);
  run<32>("v_xad_u32", out, ghz); run<33>("v_sad_u8", out, ghz); run<34>("v_lshl_add_u32", out, ghz);
  run<35>("v_add_u32_dpp wave_shr:1", out, ghz); run<36>("v_and_b32_sdwa", out, ghz);
  run<37>("v_bfi_b32", out, ghz); run<38>("v_max_u32 (VOP2)", out, ghz); run<39>("v_or3_b32", out, ghz);
  run<40>("v_and_b32_e64 (sgpr operand)", out, ghz); run<41>("v_cmp_lt_u32_e64 (to sgpr)", out, ghz); run<42>("v_add_lshl_u32", out, ghz);
  run<43>("v_lshl_add_u64", out, ghz); run<44>("v_readlane_b32 (+ s_nop 3)", out, ghz); run<45>("v_writelane_b32", out, ghz);
  run<46>("v_max_u32_dpp row_shr:1", out, ghz); run<47>("v_mov_b32 literal", out, ghz); run<48>("v_lshrrev_b64", out, ghz);
  return 0;
}
