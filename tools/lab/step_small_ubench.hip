// step_small_ubench.hip -- why does a one-cell-per-thread stencil pass take 75 us on a 4096^2 layer when a copy takes 25?
// Variants of the gather (block shape, predicated unrolled loads against a loop, 32- against 64-bit offsets), timed alone.
// hipcc --offload-arch=gfx950 -O3 tools/lab/step_small_ubench.hip -o /tmp/ssu && /tmp/ssu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

struct Win { int n; signed char di[12], dj[12]; };

template <int BX, int BY, int MODE>
__global__ __launch_bounds__(BX* BY) void k(int rows, int cols, Win w, const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= rows || j >= cols) return;
  const size_t o = (size_t)j * rows + i;
  const float c = in[o];
  float mx = c, mn = c;
  if (MODE == 0) {  // predicated, unrolled, all loads first (k_step_small)
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      v[k] = __builtin_nanf("");
      if (k < w.n) {
        const int ii = i + w.di[k], jj = j + w.dj[k];
        if ((unsigned)ii < (unsigned)rows && (unsigned)jj < (unsigned)cols) v[k] = in[(size_t)jj * rows + ii];
      }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (k < w.n && __builtin_isfinite(v[k])) { mx = fmaxf(mx, v[k]); mn = fminf(mn, v[k]); }
  } else if (MODE == 1) {  // plain loop
    for (int k = 0; k < w.n; ++k) {
      const int ii = i + w.di[k], jj = j + w.dj[k];
      if ((unsigned)ii < (unsigned)rows && (unsigned)jj < (unsigned)cols) {
        const float z = in[(size_t)jj * rows + ii];
        if (__builtin_isfinite(z)) { mx = fmaxf(mx, z); mn = fminf(mn, z); }
      }
    }
  } else if (MODE == 2) {  // 32-bit offsets, clamped indices instead of branches
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (k < w.n) {
        int ii = i + w.di[k], jj = j + w.dj[k];
        const bool ok = (unsigned)ii < (unsigned)rows && (unsigned)jj < (unsigned)cols;
        ii = ok ? ii : i; jj = ok ? jj : j;
        const float z = in[(unsigned)(jj * rows + ii)];
        if (__builtin_isfinite(z)) { mx = fmaxf(mx, z); mn = fminf(mn, z); }
      }
  }
  out[o] = __builtin_isfinite(c) ? mx - mn : __builtin_nanf("");
}

template <int BX, int BY, int MODE>
float run(int n, const Win& w, const float* in, float* out) {
  dim3 grid((n + BX - 1) / BX, (n + BY - 1) / BY), blk(BX, BY);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<BX, BY, MODE>), grid, blk, 0, 0, n, n, w, in, out);
  hipEventRecord(a);
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k<BX, BY, MODE>), grid, blk, 0, 0, n, n, w, in, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}

int main() {
  const int n = 4096;
  float *in, *out;
  hipMalloc(&in, (size_t)n * n * 4); hipMalloc(&out, (size_t)n * n * 4);
  std::vector<float> h((size_t)n * n);
  for (size_t q = 0; q < h.size(); ++q) h[q] = std::sin(0.001f * (float)(q % 100000));
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  Win w0{}; w0.n = 0;
  Win w4{}; w4.n = 4; const int d4[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  for (int k = 0; k < 4; ++k) { w4.di[k] = d4[k][0]; w4.dj[k] = d4[k][1]; }
  Win w8{}; w8.n = 8; int q = 0;
  for (int dj = -1; dj <= 1; ++dj) for (int di = -1; di <= 1; ++di) if (di || dj) { w8.di[q] = di; w8.dj[q] = dj; ++q; }
  const Win* ws[3] = {&w0, &w4, &w8};
  for (int t = 0; t < 3; ++t) {
    std::printf("n_off %d: 64x4 unrolled %.1f  64x4 loop %.1f  64x4 clamped32 %.1f | 256x1 unrolled %.1f  256x1 loop %.1f  256x1 clamped32 %.1f | 64x1 loop %.1f  64x8 clamped32 %.1f us\n", ws[t]->n,
                run<64, 4, 0>(n, *ws[t], in, out), run<64, 4, 1>(n, *ws[t], in, out), run<64, 4, 2>(n, *ws[t], in, out),
                run<256, 1, 0>(n, *ws[t], in, out), run<256, 1, 1>(n, *ws[t], in, out), run<256, 1, 2>(n, *ws[t], in, out),
                run<64, 1, 1>(n, *ws[t], in, out), run<64, 8, 2>(n, *ws[t], in, out));
  }
  return 0;
}
