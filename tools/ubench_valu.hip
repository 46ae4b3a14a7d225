// Micro-benchmark: issue rate and dependent-issue latency of the VALU ops the kernels are built from
// (gfx950).  Each kernel runs `iters` iterations of 64 instructions on CH independent accumulators;
// blocks of 64 threads, `waves` waves per SIMD resident (grid = 256 CUs * 4 SIMDs * waves).
// Output: cycles (at the nominal 2.4 GHz) per instruction per SIMD, i.e. the reciprocal issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define OPS(X)                                                                                        \
  X(0, "v_fma_f64", "v_fma_f64 %0, %0, %2, %3", d)                                                     \
  X(1, "v_add_f64", "v_add_f64 %0, %0, %2", d)                                                         \
  X(2, "v_fma_f32", "v_fma_f32 %1, %1, %4, %5", f)                                                     \
  X(3, "v_max3_f32", "v_max3_f32 %1, %1, %4, %5", f)                                                   \
  X(4, "v_max_f32", "v_max_f32 %1, %1, %4", f)                                                         \
  X(5, "v_add_u32", "v_add_u32 %1, %1, %4", f)                                                         \
  X(6, "v_add3_u32", "v_add3_u32 %1, %1, %4, %5", f)                                                   \
  X(7, "v_cndmask_vcc", "v_cndmask_b32 %1, %1, %4, vcc", f)                                            \
  X(8, "v_cndmask_sgpr", "v_cndmask_b32 %1, %1, %4, s[20:21]", f)                                      \
  X(9, "v_cmp_gt_f32", "v_cmp_gt_f32 vcc, %1, %4", f)                                                  \
  X(10, "v_cmp+cndmask", "v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc", f)                \
  X(11, "v_cmp_gt_f64", "v_cmp_gt_f64 vcc, %0, %2", d)                                                 \
  X(12, "v_and_b32", "v_and_b32 %1, %1, %4", f)                                                        \
  X(13, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 1, %4", f)                                           \
  X(14, "v_cvt_f64_f32", "v_cvt_f64_f32 %0, %1", d)                                                    \
  X(15, "v_cvt_f32_f64", "v_cvt_f32_f64 %1, %0", f)                                                    \
  X(16, "v_mov_b32", "v_mov_b32 %1, %4", f)                                                            \
  X(17, "v_readlane", "v_readlane_b32 s22, %1, 3", f)                                                  \
  X(18, "v_rsq_f64", "v_rsq_f64 %0, %0", d)                                                            \
  X(19, "v_sqrt_f32", "v_sqrt_f32 %1, %1", f)                                                          \
  X(20, "v_rcp_f64", "v_rcp_f64 %0, %0", d)                                                            \
  X(21, "v_mul_f32", "v_mul_f32 %1, %1, %4", f)                                                        \
  X(22, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %2, %3", d)                                              \
  X(23, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %2", d)                                                  \
  X(24, "v_fmac_f64", "v_fmac_f64 %0, %2, %3", d)                                                      \
  X(25, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %1, %4, %0", d)                                       \
  X(26, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 2, %2", d)                                           \
  X(27, "v_mul_lo_u32", "v_mul_lo_u32 %1, %1, %4", f)                                                  \
  X(28, "v_min3_f32", "v_min3_f32 %1, %1, %4, %5", f)                                                  \
  X(29, "v_cmp_class", "v_cmp_class_f32 vcc, %1, %4", f)                                               \
  X(30, "v_bfe_u32", "v_bfe_u32 %1, %1, 3, 5", f)                                                      \
  X(31, "v_med3_f32", "v_med3_f32 %1, %1, %4, %5", f)                                                  \
  X(32, "v_cvt_f64_i32", "v_cvt_f64_i32 %0, %1", d)                                                    \
  X(33, "v_fma_f64_sgpr", "v_fma_f64 %0, %0, s[20:21], %3", d)                                         \
  X(34, "v_max_f64", "v_max_f64 %0, %0, %2", d)                                                        \
  X(35, "v_mul_f64", "v_mul_f64 %0, %0, %2", d)                                                        \
  X(36, "v_ldexp_f64", "v_ldexp_f64 %0, %0, %4", d)                                                    \
  X(37, "v_floor_f64", "v_floor_f64 %0, %0", d)                                                        \
  X(38, "s_nop(valu-free)", "s_nop 0", f)                                                              \
  X(39, "v_add_f32", "v_add_f32 %1, %1, %4", f)                                                        \
  X(40, "v_sub_u32", "v_sub_u32 %1, %1, %4", f)                                                        \
  X(41, "v_or_b32", "v_or_b32 %1, %1, %4", f)                                                          \
  X(42, "v_xor+cmp pair", "v_cmp_lt_u32 vcc, %1, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc", f)                 \
  X(43, "v_max_u32", "v_max_u32 %1, %1, %4", f)                                                        \
  X(44, "v_max_i32", "v_max_i32 %1, %1, %4", f)                                                        \
  X(45, "v_min_u32", "v_min_u32 %1, %1, %4", f)                                                        \
  X(46, "v_pk_max_i16", "v_pk_max_i16 %1, %1, %4", f)                                                  \
  X(47, "v_pk_add_u16", "v_pk_add_u16 %1, %1, %4", f)                                                  \
  X(48, "v_add_u32_sdwa", "v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3", f) \
  X(49, "v_lshl_or_b32", "v_lshl_or_b32 %1, %1, 3, %4", f)                                             \
  X(50, "v_cvt_u32_f32", "v_cvt_u32_f32 %1, %1", f)                                                    \
  X(51, "v_cvt_f32_u32", "v_cvt_f32_u32 %1, %1", f)                                                    \
  X(52, "v_max3_u32", "v_max3_u32 %1, %1, %4, %5", f)                                                  \
  X(53, "v_pk_max_f16", "v_pk_max_f16 %1, %1, %4", f)                                                  \
  X(54, "v_maximum3_f32?", "v_max3_f32 %1, %1, %4, %4", f)

template <int OP, int CH>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double a[CH];
  float f[CH];
  for (int c = 0; c < CH; ++c) { a[c] = seed + c + threadIdx.x; f[c] = (float)a[c]; }
  const double m = seed * 0.5, n = seed * 0.25;
  const float mf = (float)m, nf = (float)n;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#define X(ID, NAME, ASM, KIND) \
  if (OP == ID) asm volatile(ASM : "+v"(a[c]), "+v"(f[c]) : "v"(m), "v"(n), "v"(mf), "v"(nf) : "vcc", "s20", "s21", "s22");
        OPS(X)
#undef X
      }
    }
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += a[c] + f[c];
  if (s == 12345.678) out[0] = s;
}

template <int OP, int CH>
double run(int waves) {
  double* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  const int grid = 256 * 4 * waves;
  hipLaunchKernelGGL((k<OP, CH>), dim3(grid), dim3(64), 0, 0, d, 100, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CH>), dim3(grid), dim3(64), 0, 0, d, iters, 1.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms * 1e-3 * 2.4e9 / (64.0 * iters * waves);
}

int main(int argc, char** argv) {
  const int first = argc > 1 ? atoi(argv[1]) : 0;  // print the ops from this id on
  printf("%-18s %10s %10s %10s %10s %10s %10s   (cycles per instruction per SIMD; dep = 1 dependent chain, 1 wave)\n", "op", "dep,1w",
         "8ch,1w", "8ch,2w", "8ch,3w", "8ch,4w", "8ch,8w");
#define X(ID, NAME, ASM, KIND) \
  if (ID >= first)             \
    printf("%-18s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", NAME, run<ID, 1>(1), run<ID, 8>(1), run<ID, 8>(2), run<ID, 8>(3), run<ID, 8>(4), run<ID, 8>(8));
  OPS(X)
#undef X
  return 0;
}
