#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 20000
#define U 16
__global__ void dep_add(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U; u++) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x)); }
  }
  out[threadIdx.x] = a;
}
__global__ void indep_add4(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x, b = 1, c = 2, d = 3;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U / 4; u++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(x));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(x));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(x));
    }
  }
  out[threadIdx.x] = a + b + c + d;
}
__global__ void dep_dpp(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U; u++) { asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a)); }
  }
  out[threadIdx.x] = a;
}
__global__ void dep_mul(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U; u++) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x)); }
  }
  out[threadIdx.x] = a;
}
__global__ void indep_mul4(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x, b = 1, c = 2, d = 3;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U / 4; u++) {
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d) : "v"(x));
    }
  }
  out[threadIdx.x] = a + b + c + d;
}
// one dependent add followed by K independent adds per step
template <int K>
__global__ void dep_plus_indep(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x, b[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#pragma unroll
      for (int k = 0; k < K; k++) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[k]) : "v"(x)); }
    }
  }
  uint32_t s = a; for (int k = 0; k < 8; k++) s += b[k];
  out[threadIdx.x] = s;
}
__global__ void dot_mad(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x, c = 0, h = threadIdx.x * 2654435761u;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U / 2; u++) {
      asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(h));
      asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(h));
    }
  }
  out[threadIdx.x] = a + (c << 16);
}
// low 32 bits of k*v + c by v_mad_u64_u32 (the high half is thrown away): is it issued like v_mul_lo_u32 + v_add_u32, or slower?
__global__ void mad64_indep4(uint32_t* out, uint32_t x) {
  unsigned long long a = threadIdx.x, b = 1, c = 2, d = 3;
  const unsigned long long k = 16384;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U / 4; u++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(a) : "v"(x), "v"((uint32_t)a), "v"(k) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(b) : "v"(x), "v"((uint32_t)b), "v"(k) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(c) : "v"(x), "v"((uint32_t)c), "v"(k) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(x), "v"((uint32_t)d), "v"(k) : "vcc");
    }
  }
  out[threadIdx.x] = (uint32_t)(a + b + c + d);
}
__global__ void muladd_indep4(uint32_t* out, uint32_t x) {
  uint32_t a = threadIdx.x, b = 1, c = 2, d = 3;
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = 0; u < U / 4; u++) {
      asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %0, 0x4000, %0" : "+v"(a) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %0, 0x4000, %0" : "+v"(b) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %0, 0x4000, %0" : "+v"(c) : "v"(x));
      asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %0, 0x4000, %0" : "+v"(d) : "v"(x));
    }
  }
  out[threadIdx.x] = a + b + c + d;
}
__global__ void clock_probe(unsigned long long* out) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t a = threadIdx.x;
  for (int i = 0; i < N * 4; i++) { asm volatile("v_add_u32 %0, %0, %0" : "+v"(a)); }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = a; }
}
template <typename F> static float run(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  uint32_t* d; hipMalloc(&d, 1 << 20);
  const double per = 1e6 / ((double)N * U);
  for (int w = 1; w <= 2; w++) {
    printf("--- %d wave(s) on one SIMD (block of %d threads)\n", w, 64 * (w == 1 ? 1 : 8));
    dim3 blk(w == 1 ? 64 : 512);   // 512 threads = 8 waves = 2 per SIMD
    printf("dep v_add_u32            : %.2f ns per instr\n", run([&] { hipLaunchKernelGGL(dep_add, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("4 indep v_add_u32 chains : %.2f ns per instr\n", run([&] { hipLaunchKernelGGL(indep_add4, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("dep s_nop1 + v_add_dpp   : %.2f ns per pair\n", run([&] { hipLaunchKernelGGL(dep_dpp, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("dep v_mul_lo_u32         : %.2f ns per instr\n", run([&] { hipLaunchKernelGGL(dep_mul, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("4 indep v_mul_lo_u32     : %.2f ns per instr\n", run([&] { hipLaunchKernelGGL(indep_mul4, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("1 dep + 1 indep add      : %.2f ns per step\n", run([&] { hipLaunchKernelGGL(dep_plus_indep<1>, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("1 dep + 2 indep add      : %.2f ns per step\n", run([&] { hipLaunchKernelGGL(dep_plus_indep<2>, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("1 dep + 4 indep add      : %.2f ns per step\n", run([&] { hipLaunchKernelGGL(dep_plus_indep<4>, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("4 indep v_mad_u64_u32    : %.2f ns per instr\n", run([&] { hipLaunchKernelGGL(mad64_indep4, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("4 indep mul_lo + add     : %.2f ns per pair\n", run([&] { hipLaunchKernelGGL(muladd_indep4, dim3(1), blk, 0, 0, d, 3u); }) * per);
    printf("dot2_u16 + mad_u32_u16   : %.2f ns per pair\n", run([&] { hipLaunchKernelGGL(dot_mad, dim3(1), blk, 0, 0, d, 0x00030005u); }) * per * 2);
  }
  unsigned long long* q; hipMalloc(&q, 64); unsigned long long h[3];
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, q); hipMemcpy(h, q, 24, hipMemcpyDeviceToHost);
    printf("clock: %llu shader ticks in %llu x 10 ns -> %.0f MHz\n", h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
  }
  // correctness of the 16-bit product: compare with a 32-bit multiply on a few values
  return 0;
}
