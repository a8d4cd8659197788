// Sustained VALU cost of the PARCOR lattice's instruction mix (src/SLAPredictor.c:557-607) on a full chip, in registers only:
// what k_lattice_groups could reach if nothing but its arithmetic counted.  One lane keeps LAT_T = 16 consecutive forward /
// backward errors, a stage is 2 terms R(k*v) = (k*v + 2^14) >> 15 and 2 subtractions per sample, the previous lane's last
// backward error arrives by a DPP wave shift.  Variants of the term:
//   0  v_mul_lo_u32, v_add_u32, v_ashrrev_i32, v_sub_u32            (the kernel of round 3: 8 lane-ops per sample and stage)
//   1  v_mad_i32_i24(k, v, 2^14), v_ashrrev_i32, v_sub_u32          (|v| < 2^23: the low 32 bits are the reference's wrapped ones)
//   2  v_mad_i32_i24(2k, v, 2^15), v_sub_u32_sdwa sext(WORD_1)      (|v| < 2^23 and |k*v| + 2^14 < 2^30: no wrap to reproduce)
//   3  v_mul_lo_u32(2k, v), v_add_u32 2^15, v_sub_u32_sdwa          (any v, |k*v| + 2^14 < 2^30)
//   4  v_mad_i64_i32(k << 17, v, 2^31) high dword, v_sub_u32        (|k| < 2^14, |k*v| + 2^14 < 2^31... no wrap)
//   5  v_mul_hi_i32 only + v_sub_u32                                 (NOT the lattice: the price of a mul_hi)
//   6  adds only, 8 per sample and stage                             (the issue peak itself)
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_lattice ubench_lattice.hip ; run: ./ubench_lattice [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define LAT_T 16
#define STAGES 32
#define REPS 64

// one sample of one stage: nb = B - R(k F), F = F - R(k B); the two products first, then the two subtractions (a result
// is never read by the very next instruction: no wait states for the SDWA / DPP readers)
#define SDWA_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
// one sample of one stage: B[j] = B[i] - R(k F[i]), F[i] = F[i] - R(k B[i])   (j = i + 1; both products first, then both subtractions)
#define STEP_V0(i, j) "v_mul_lo_u32 %[t1], %[k], %[F" #i "]\n v_mul_lo_u32 %[t2], %[k], %[B" #i "]\n v_add_u32 %[t1], 0x4000, %[t1]\n v_add_u32 %[t2], 0x4000, %[t2]\n" \
                      "v_ashrrev_i32 %[t1], 15, %[t1]\n v_ashrrev_i32 %[t2], 15, %[t2]\n v_sub_u32 %[B" #j "], %[B" #i "], %[t1]\n v_sub_u32 %[F" #i "], %[F" #i "], %[t2]\n"
#define STEP_V1(i, j) "v_mad_i32_i24 %[t1], %[k], %[F" #i "], %[c]\n v_mad_i32_i24 %[t2], %[k], %[B" #i "], %[c]\n v_ashrrev_i32 %[t1], 15, %[t1]\n v_ashrrev_i32 %[t2], 15, %[t2]\n" \
                      "v_sub_u32 %[B" #j "], %[B" #i "], %[t1]\n v_sub_u32 %[F" #i "], %[F" #i "], %[t2]\n"
#define STEP_V2(i, j) "v_mad_i32_i24 %[t1], %[k], %[F" #i "], %[c]\n v_mad_i32_i24 %[t2], %[k], %[B" #i "], %[c]\n" \
                      "v_sub_u32_sdwa %[B" #j "], %[B" #i "], sext(%[t1])" SDWA_HI "v_sub_u32_sdwa %[F" #i "], %[F" #i "], sext(%[t2])" SDWA_HI
#define STEP_V3(i, j) "v_mul_lo_u32 %[t1], %[k], %[F" #i "]\n v_mul_lo_u32 %[t2], %[k], %[B" #i "]\n v_add_u32 %[t1], 0x8000, %[t1]\n v_add_u32 %[t2], 0x8000, %[t2]\n" \
                      "v_sub_u32_sdwa %[B" #j "], %[B" #i "], sext(%[t1])" SDWA_HI "v_sub_u32_sdwa %[F" #i "], %[F" #i "], sext(%[t2])" SDWA_HI
#define STEP_V4(i, j) "v_mad_i64_i32 v[60:61], vcc, %[k], %[F" #i "], %[c]\n v_mad_i64_i32 v[62:63], vcc, %[k], %[B" #i "], %[c]\n" \
                      "v_sub_u32 %[B" #j "], %[B" #i "], v61\n v_sub_u32 %[F" #i "], %[F" #i "], v63\n"
#define STEP_V5(i, j) "v_mul_hi_i32 %[t1], %[k], %[F" #i "]\n v_mul_hi_i32 %[t2], %[k], %[B" #i "]\n v_sub_u32 %[B" #j "], %[B" #i "], %[t1]\n v_sub_u32 %[F" #i "], %[F" #i "], %[t2]\n"
#define STEP_V6(i, j) "v_add_u32 %[t1], %[k], %[F" #i "]\n v_add_u32 %[t2], %[k], %[B" #i "]\n v_add_u32 %[t1], %[t1], %[F" #i "]\n v_add_u32 %[t2], %[t2], %[B" #i "]\n" \
                      "v_add_u32 %[t1], %[t1], %[F" #i "]\n v_add_u32 %[t2], %[t2], %[B" #i "]\n v_sub_u32 %[B" #j "], %[B" #i "], %[t1]\n v_sub_u32 %[F" #i "], %[F" #i "], %[t2]\n"
#define STAGE(S) S(15, 16) S(14, 15) S(13, 14) S(12, 13) S(11, 12) S(10, 11) S(9, 10) S(8, 9) S(7, 8) S(6, 7) S(5, 6) S(4, 5) S(3, 4) S(2, 3) S(1, 2) S(0, 1)
#define FB_OPERANDS [F0] "+v"(F[0]), [F1] "+v"(F[1]), [F2] "+v"(F[2]), [F3] "+v"(F[3]), [F4] "+v"(F[4]), [F5] "+v"(F[5]), [F6] "+v"(F[6]), [F7] "+v"(F[7]), \
                    [F8] "+v"(F[8]), [F9] "+v"(F[9]), [F10] "+v"(F[10]), [F11] "+v"(F[11]), [F12] "+v"(F[12]), [F13] "+v"(F[13]), [F14] "+v"(F[14]), [F15] "+v"(F[15]), \
                    [B0] "+v"(B[0]), [B1] "+v"(B[1]), [B2] "+v"(B[2]), [B3] "+v"(B[3]), [B4] "+v"(B[4]), [B5] "+v"(B[5]), [B6] "+v"(B[6]), [B7] "+v"(B[7]), [B8] "+v"(B[8]), \
                    [B9] "+v"(B[9]), [B10] "+v"(B[10]), [B11] "+v"(B[11]), [B12] "+v"(B[12]), [B13] "+v"(B[13]), [B14] "+v"(B[14]), [B15] "+v"(B[15]), [B16] "+v"(B[16])

template <int V>
__global__ __launch_bounds__(256) void lat(int32_t* out, const int32_t* __restrict__ kc, int32_t seed)
{
  // F[i] = f[n0 + i], B[i] = b[n0 + i - 1] (the backward error one sample late, as the stage wants it); a stage walks i downwards and
  // writes the new backward error of sample n0 + i into B[i + 1], whose old value the step before has used up: no register moves
  int32_t F[LAT_T], B[LAT_T + 1];
#pragma unroll
  for (int i = 0; i < LAT_T; i++) { F[i] = (int32_t)(threadIdx.x * 37u + i * seed) >> 12; B[i + 1] = F[i]; }
  B[0] = 0;
  for (int rep = 0; rep < REPS; rep++) {
    for (int m = 1; m <= STAGES; m++) {
      const int32_t ks = __builtin_amdgcn_readfirstlane(kc[m]);
      int32_t t1, t2;
      if (V == 4) {
        const int32_t k = ks << 17;      // the 64-bit results live in v[60:63]: inline asm cannot name the high half of an operand
        asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V4)
                     : FB_OPERANDS : [k] "v"(k), [c] "v"(0x80000000ll) : "vcc", "v60", "v61", "v62", "v63");
      } else {
        const int32_t k = (V == 2 || V == 3) ? ks * 2 : (V == 5) ? (ks << 17) : ks;
        const int32_t c = (V == 2) ? 0x8000 : 0x4000;
        if (V == 0) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V0) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
        if (V == 1) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V1) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
        if (V == 2) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V2) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
        if (V == 3) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V3) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
        if (V == 5) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V5) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
        if (V == 6) { asm volatile("s_nop 1\n v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" STAGE(STEP_V6) : FB_OPERANDS, [t1] "=&v"(t1), [t2] "=&v"(t2) : [k] "v"(k), [c] "s"(c)); }
      }
    }
  }
  int32_t s = 0;
#pragma unroll
  for (int i = 0; i < LAT_T; i++) { s += F[i] ^ B[i + 1]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V> static double run(int32_t* d, const int32_t* kc, int blocks)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(lat<V>, dim3(blocks), dim3(256), 0, 0, d, kc, 12345); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) { hipLaunchKernelGGL(lat<V>, dim3(blocks), dim3(256), 0, 0, d, kc, 12345); }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5.0;
}

int main(int argc, char** argv)
{
  const int wps = argc > 1 ? atoi(argv[1]) : 8;          // waves per SIMD
  const int blocks = 256 * wps;                            // 256 CUs x (wps x 4 SIMDs / 4 waves per block)
  int32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  int32_t hk[STAGES + 1]; for (int i = 0; i <= STAGES; i++) { hk[i] = (i * 7919) % 97 - 48; }
  int32_t* kc; hipMalloc(&kc, sizeof(hk)); hipMemcpy(kc, hk, sizeof(hk), hipMemcpyHostToDevice);
  const double terms = (double)blocks * 256.0 * LAT_T * STAGES * REPS * 2.0;         // R(k*v) terms per launch
  const char* name[7] = {"mul_lo+add+ashr+sub (round 3)", "mad_i32_i24+ashr+sub", "mad_i32_i24(2k)+sub_sdwa", "mul_lo(2k)+add+sub_sdwa",
                         "mad_i64_i32 hi+sub", "mul_hi_i32+sub (not the lattice)", "3 adds + sub per term (issue peak)"};
  const int ops[7] = {4, 3, 2, 3, 2, 2, 4};
  double ms[7];
  ms[0] = run<0>(d, kc, blocks); ms[1] = run<1>(d, kc, blocks); ms[2] = run<2>(d, kc, blocks); ms[3] = run<3>(d, kc, blocks);
  ms[4] = run<4>(d, kc, blocks); ms[5] = run<5>(d, kc, blocks); ms[6] = run<6>(d, kc, blocks);
  printf("%d waves per SIMD, %d blocks of 256; peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s\n", wps, blocks);
  for (int v = 0; v < 7; v++) {
    printf("variant %d  %-34s %8.3f ms  %7.2f G terms/s  %6.2f T instr-lanes/s (%d per term)  %5.1f ps per term-lane\n", v, name[v], ms[v],
           terms / ms[v] / 1e6, terms * ops[v] / ms[v] / 1e9, ops[v], ms[v] * 1e9 / terms);
  }
  return 0;
}
