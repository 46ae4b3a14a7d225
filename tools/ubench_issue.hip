// Micro-benchmark: how instructions of different types share a wave's issue slots (gfx950).
// Loop body = 32 x { v_fma_f64 on one of 8 independent accumulators ; <companion instruction> }.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double a[8];
  for (int c = 0; c < 8; ++c) a[c] = seed + c + threadIdx.x;
  const double m = seed * 0.5, n = seed * 0.25;
  int s0 = iters, s1 = 3;
  __shared__ double lds[256];
  lds[threadIdx.x] = seed;
  const double* lp = lds + threadIdx.x;
  double l0 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u % 8]) : "v"(m), "v"(n));
      if (MODE == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
      if (MODE == 2) asm volatile("s_nop 0");
      if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(u + 4) % 8]) : "v"(m), "v"(n));
      if (MODE == 4) asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"((unsigned)(size_t)lp));
      if (MODE == 5) asm volatile("s_cmp_lt_i32 %0, %1\n s_cselect_b32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
    }
    if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  double s = l0 + s0;
  for (int c = 0; c < 8; ++c) s += a[c];
  if (s == 12345.678) out[0] = s;
}

template <int MODE>
void run(const char* name) {
  double* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-34s", name);
  for (int waves = 1; waves <= 4; ++waves) {
    const int iters = 8000, grid = 256 * 4 * waves;
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, 100, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %dw: %6.2f", waves, ms * 1e-3 * 2.4e9 / (32.0 * iters * waves));
  }
  printf("   cycles per {pair} per SIMD\n");
  hipFree(d);
}

int main() {
  run<0>("v_fma_f64 alone");
  run<3>("v_fma_f64 + v_fma_f64");
  run<1>("v_fma_f64 + s_add_u32");
  run<5>("v_fma_f64 + s_cmp + s_cselect");
  run<2>("v_fma_f64 + s_nop");
  run<4>("v_fma_f64 + ds_read_b64");
  return 0;
}
