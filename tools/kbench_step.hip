// Development micro-benchmark: times variants of the marching step kernel (not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <math.h>
#include "te_march.h"
using namespace te; using namespace te::fast;

template <int Q, int VAR>
__global__ __launch_bounds__(kLanes) void kvar(Geo g, const float* __restrict__ elev, float* __restrict__ sh, Region rg) {
  using S = Shape<Q>; using T = Strip<Q>;
  constexpr int R = S::R, P = S::P, W = T::W;
  __shared__ float rowbuf[P * W];
  const int lane = threadIdx.x;
  const size_t mo = 0;
  const int i0 = rg.i0 + blockIdx.x * kLanes;
  const int js = rg.j0 + blockIdx.y * T::out_rows;
  const int i = i0 + lane;
  float amax[P], amin[P], zc[P];
#pragma unroll
  for (int k = 0; k < P; ++k) amax[k] = amin[k] = zc[k] = qnan();
  float stage[T::NLD];
  if (VAR != 1) load_period<Q>(stage, elev + mo, g, js - R, i0 - R, lane);
  else { for (int k = 0; k < T::NLD; ++k) stage[k] = lane * 0.01f + k; }
#pragma unroll 1
  for (int per = 0; per < T::periods; ++per) {
    const int rbase = js - R + per * P;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < T::NLD; ++k) {
      const int idx = lane + k * kLanes;
      const float t = stage[k];
      if (idx < P * W) rowbuf[idx] = __builtin_isfinite(t) ? t : qnan();
    }
    __syncthreads();
    if (VAR != 1) { if (per + 1 < T::periods) load_period<Q>(stage, elev + mo, g, rbase + P, i0 - R, lane); }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float* row = rowbuf + p * W + lane + R;
      float mx[R + 1], mn[R + 1];
      mx[0] = mn[0] = row[0];
#pragma unroll
      for (int d = 1; d <= R; ++d) {
        const float a = row[-d], b = row[d];
        if (VAR == 3) { mx[d] = fmaxf(mx[d-1], fmaxf(a, b)); mn[d] = fminf(mn[d-1], fminf(a, b)); }
        else { mx[d] = vmax3(mx[d - 1], a, b); mn[d] = vmin3(mn[d - 1], a, b); }
      }
      zc[p] = row[0];
#pragma unroll
      for (int e = -R; e <= R; ++e) {
        const int slot = (p + e + P) % P;
        const int w = S::hw(e < 0 ? -e : e);
        if (VAR == 3) { amax[slot] = fmaxf(amax[slot], mx[w]); amin[slot] = fminf(amin[slot], mn[w]); }
        else { amax[slot] = vmax2(amax[slot], mx[w]); amin[slot] = vmin2(amin[slot], mn[w]); }
      }
      const int so = (p + R + 1) % P;
      const int j = rbase + p - R;
      if (VAR != 2) {
        if (j >= js && j < js + T::out_rows && j < rg.j1 && i < rg.i1) {
          const float z0 = zc[so];
          const float out = (z0 == z0) ? (float)((double)amax[so] - (double)amin[so]) : qnan();
          sh[mo + (size_t)j * g.rows + i] = out;
        }
      } else {
        if (amax[so] == 12345.f) sh[0] = amin[so];
      }
      amax[so] = amin[so] = qnan();
    }
  }
}

static int g_cold = 0;
template <int Q, int VAR> float run(Geo g, const float* e, float* o, int iters) {
  using T = Strip<Q>;
  Region r{-1, 0, 0, g.rows, g.cols};
  dim3 grid((g.rows + 63) / 64, (g.cols + T::out_rows - 1) / T::out_rows, 1);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((kvar<Q, VAR>), grid, dim3(64), 0, 0, g, e, o, r);
  static char* thrash = nullptr;
  if (!thrash) hipMalloc(&thrash, (size_t)1 << 30);
  float tot = 0;
  for (int k = 0; k < iters; ++k) {
    if (g_cold) hipMemsetAsync(thrash, k, (size_t)1 << 30, 0);
    hipEventRecord(a);
    hipLaunchKernelGGL((kvar<Q, VAR>), grid, dim3(64), 0, 0, g, e, o, r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
  }
  return tot / iters * 1000.f;
}

float run_lib(Geo g, const float* e, float* o, int iters) {
  Region r{-1, 0, 0, g.rows, g.cols};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float tot = 0;
  for (int k = 0; k < iters + 2; ++k) {
    hipEventRecord(a);
    te::fast::step_height_fast(81, g, e, o, r, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (k >= 2) tot += ms;
  }
  return tot / iters * 1000.f;
}
int main() {
  const int n = 4096;
  Geo g{}; g.rows = n; g.cols = n; g.batch = 1; g.res = 0.05;
  std::vector<float> h((size_t)n * n);
  for (size_t k = 0; k < h.size(); ++k) h[k] = (float)((k * 2654435761u) % 1000) * 1e-3f;
  float *e, *o; hipMalloc(&e, h.size() * 4); hipMalloc(&o, h.size() * 4);
  hipMemcpy(e, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("lib kernel Q81: %.1f us\n", run_lib(g, e, o, 10));
  for (size_t k = 0; k < h.size(); ++k) h[k] = 0.3f * sinf(0.01f * (k % 4096)) * cosf(0.013f * (k / 4096));
  hipMemcpy(e, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("lib kernel Q81 smooth data: %.1f us\n", run_lib(g, e, o, 10));
  for (g_cold = 0; g_cold < 1; ++g_cold) {
  printf("cold=%d\n", g_cold);
  printf("Q81 full      : %.1f us\n", run<81, 0>(g, e, o, 10));
  printf("Q81 no gloads : %.1f us\n", run<81, 1>(g, e, o, 10));
  printf("Q81 no stores : %.1f us\n", run<81, 2>(g, e, o, 10));
  printf("Q81 fmaxf     : %.1f us\n", run<81, 3>(g, e, o, 10));
  printf("Q25 full      : %.1f us\n", run<25, 0>(g, e, o, 10));
  printf("Q25 no gloads : %.1f us\n", run<25, 1>(g, e, o, 10));
  }
  return 0;
}
