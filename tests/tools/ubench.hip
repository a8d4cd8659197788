// micro-benchmarks that calibrate the latency model used in DESIGN.md (dev tool; build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N 100000

__global__ void dep_add(double* out, double x, int chains)
{
  double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0;
  if (chains == 1) { for (int i = 0; i < N; i++) { a0 += x; } }
  else if (chains == 2) { for (int i = 0; i < N; i++) { a0 += x; a1 += x; } }
  else { for (int i = 0; i < N; i++) { a0 += x; a1 += x; a2 += x; a3 += x; } }
  out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3;
}

__global__ void dep_addmul(double* out, double x)
{
  double a0 = threadIdx.x;
  for (int i = 0; i < N; i++) { a0 += x * x; x += 1e-9; }     // one independent mul + add per dependent add
  out[threadIdx.x + blockIdx.x * blockDim.x] = a0;
}

__global__ void dep_iadd(uint32_t* out, uint32_t x)
{
  uint32_t a0 = threadIdx.x;
  for (int i = 0; i < N; i++) { a0 = a0 * x + 1; }
  out[threadIdx.x + blockIdx.x * blockDim.x] = a0;
}

__global__ void lds_chase(uint32_t* out, int stride)
{
  __shared__ uint32_t tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) { tab[i] = (i + stride) & 4095; }
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (int i = 0; i < N; i++) { p = tab[p]; }
  out[threadIdx.x + blockIdx.x * blockDim.x] = p;
}

// independent LDS reads, one wave (or more): throughput per wave-instruction
template <int WIDTH>
__global__ void lds_stream(double* out, int active)
{
  __shared__ double tab[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) { tab[i] = i; }
  __syncthreads();
  double acc = 0.0;
  if ((int)(threadIdx.x & 63) < active) {
    const int base = (threadIdx.x & 63) * WIDTH;
    for (int i = 0; i < N / 16; i++) {
      const int o = (i & 7) * 512;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (WIDTH == 1) { acc += tab[o + base + u * 64]; }
        else { const double2 v = *(const double2*)&tab[o + base + u * 128]; acc += v.x; acc += v.y; }
      }
    }
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
}

template <typename F> static float run(F f)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
  double* d; hipMalloc(&d, 1 << 20);
  for (int waves = 1; waves <= 4; waves *= 2) {
    for (int chains = 1; chains <= 4; chains *= 2) {
      float ms = run([&] { hipLaunchKernelGGL(dep_add, dim3(1), dim3(64 * waves * 4), 0, 0, d, 1.5, chains); });
      printf("dep v_add_f64: %d wave(s)/SIMD, %d chain(s)/lane: %.2f ns per add-step\n", waves, chains, ms * 1e6 / N);
    }
  }
  for (int grid = 1; grid <= 2048; grid *= 8) {
    float ms = run([&] { hipLaunchKernelGGL(dep_add, dim3(grid), dim3(256), 0, 0, d, 1.5, 1); });
    printf("dep v_add_f64: grid %d x 256 threads, 1 chain: %.2f ns per add-step\n", grid, ms * 1e6 / N);
    ms = run([&] { hipLaunchKernelGGL(dep_add, dim3(grid), dim3(64), 0, 0, d, 1.5, 4); });
    printf("v_add_f64 x4 indep: grid %d x 64 threads: %.2f ns per 4 adds\n", grid, ms * 1e6 / N);
  }
  for (int active = 4; active <= 64; active *= 4) {
    float ms = run([&] { hipLaunchKernelGGL(lds_stream<1>, dim3(1), dim3(64), 0, 0, d, active); });
    printf("ds_read_b64 + dep add, %d lanes, 1 wave: %.2f ns per load\n", active, ms * 1e6 / N);
    ms = run([&] { hipLaunchKernelGGL(lds_stream<2>, dim3(1), dim3(64), 0, 0, d, active); });
    printf("ds_read_b128 + 2 dep adds, %d lanes, 1 wave: %.2f ns per load (2 values)\n", active, ms * 1e6 / N);
  }
  { float ms = run([&] { hipLaunchKernelGGL(lds_stream<1>, dim3(1), dim3(256), 0, 0, d, 64); }); printf("ds_read_b64 + dep add, 4 waves (1/SIMD): %.2f ns per load per wave\n", ms * 1e6 / N); }
  { float ms = run([&] { hipLaunchKernelGGL(dep_addmul, dim3(1), dim3(64), 0, 0, d, 1.5); }); printf("dep add + indep mul + indep add: %.2f ns per step\n", ms * 1e6 / N); }
  { float ms = run([&] { hipLaunchKernelGGL(dep_iadd, dim3(1), dim3(64), 0, 0, (uint32_t*)d, 3u); }); printf("dep v_mad_u32: %.2f ns per step\n", ms * 1e6 / N); }
  { float ms = run([&] { hipLaunchKernelGGL(lds_chase, dim3(1), dim3(64), 0, 0, (uint32_t*)d, 1); }); printf("LDS pointer chase (no conflicts): %.2f ns per load\n", ms * 1e6 / N); }
  { float ms = run([&] { hipLaunchKernelGGL(lds_chase, dim3(1), dim3(64), 0, 0, (uint32_t*)d, 64); }); printf("LDS pointer chase (same-bank stride): %.2f ns per load\n", ms * 1e6 / N); }
  return 0;
}
