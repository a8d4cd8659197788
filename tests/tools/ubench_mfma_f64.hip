// v_mfma_f64_16x16x4_f64 on gfx950: issue rate (chip-wide, 1 / 2 / 4 waves per SIMD, independent accumulators) and the operand /
// result layout, found by feeding unit matrices.  Round 4: is a Hankel-structured FP64 contraction (the autocorrelation
// r[lag] = sum x[n] x[n+lag], any summation order, FMA allowed) cheaper on the matrix pipe than as v_fma_f64 with partner
// shuffles?  build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_f64 ubench_mfma_f64.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void rate(double* out, int reps)
{
  d4 acc[NACC];
  for (int a = 0; a < NACC; a++) { acc[a] = (d4){0.0, 0.0, 0.0, 0.0}; }
  double x = (double)threadIdx.x * 1e-3, y = 1.0 + (double)(threadIdx.x & 15) * 1e-4;
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int a = 0; a < NACC; a++) { acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0); }
  }
  double s = 0.0;
  for (int a = 0; a < NACC; a++) { s += acc[a].x + acc[a].y + acc[a].z + acc[a].w; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_rate(double* out, int reps)
{
  double a[8];
  for (int i = 0; i < 8; i++) { a[i] = (double)threadIdx.x + i; }
  const double x = 1.0000001, y = 1e-9;
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = __builtin_fma(a[i], x, y); }
  }
  double s = 0.0;
  for (int i = 0; i < 8; i++) { s += a[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// layout: A = e_(ia, ka) (1 at row ia, k index ka), B = e_(kb, jb): D has a 1 at (ia, jb) iff ka == kb
__global__ void layout(double* out)
{
  const int l = threadIdx.x;
  // hypothesis: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; D register r of lane l = D[4 * (l / 16) + r][l % 16]
  for (int t = 0; t < 4; t++) {
    const int ia = 3 + t, ka = t, jb = 9 - t, kb = t;
    const double a = (l % 16 == ia && l / 16 == ka) ? 1.0 : 0.0;
    const double b = (l / 16 == kb && l % 16 == jb) ? 1.0 : 0.0;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[(t * 64 + l) * 4 + 0] = acc.x; out[(t * 64 + l) * 4 + 1] = acc.y; out[(t * 64 + l) * 4 + 2] = acc.z; out[(t * 64 + l) * 4 + 3] = acc.w;
  }
}

int main()
{
  double* d; hipMalloc(&d, 8u << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 4000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;
    float ms;
    hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 0, 0, d, reps); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 0, 0, d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * reps * 4;        // wave-instructions
    printf("%d wave(s) per SIMD: v_mfma_f64_16x16x4_f64 x 4 accumulators: %.3f ms, %.1f TFLOP/s, %.1f ns per instruction and SIMD (x2.4 GHz = %.0f cycles)\n",
           wps, ms, n * 2048.0 / ms / 1e9, ms * 1e6 / (n / 1024.0), ms * 1e6 / (n / 1024.0) * 2.4);
    hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, reps); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double n1 = (double)blocks * 4 * reps;
    printf("%d wave(s) per SIMD: one dependent accumulator: %.1f TFLOP/s, %.1f ns per instruction\n", wps, n1 * 2048.0 / ms / 1e9, ms * 1e6 / (n1 / 1024.0));
    hipLaunchKernelGGL(fma_rate, dim3(blocks), dim3(256), 0, 0, d, reps); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(fma_rate, dim3(blocks), dim3(256), 0, 0, d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d wave(s) per SIMD: v_fma_f64 x 8 independent: %.1f TFLOP/s\n", wps, (double)blocks * 256 * reps * 8 * 2 / ms / 1e9);
  }
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
  static double h[4 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int t = 0; t < 4; t++) {
    const int ia = 3 + t, jb = 9 - t;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      const double want = (4 * (l / 16) + r == ia && l % 16 == jb) ? 1.0 : 0.0;
      if (h[(t * 64 + l) * 4 + r] != want) { ok = 0; printf("layout: test %d lane %d reg %d = %g, hypothesis says %g\n", t, l, r, h[(t * 64 + l) * 4 + r], want); }
    }
  }
  printf("layout hypothesis (A[l%%16][l/16], B[l/16][l%%16], D reg r = D[4(l/16)+r][l%%16]): %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
