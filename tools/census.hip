// census.hip -- how many single-wave workgroups of a given LDS / register footprint are resident per CU at once?
// Every wave records its hardware ids and its start / end time (s_memtime); the host counts the largest number of
// waves alive at the same instant on one CU.  hipcc --offload-arch=gfx950 -O2 tools/census.hip -o tools/census
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Rec {
  unsigned hw_id, xcc_id;
  unsigned long long t0, t1;
};

template <int LDS_BYTES, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_census(Rec* out, int spin) {
  __shared__ double buf[LDS_BYTES / 8];
  const unsigned long long t0 = __builtin_readcyclecounter();
  double acc = threadIdx.x;
  for (int k = threadIdx.x; k < LDS_BYTES / 8; k += 64) buf[k] = acc;
  __syncthreads();
  for (int it = 0; it < spin; ++it) {
    acc = acc * 1.0000001 + buf[(threadIdx.x + it) % (LDS_BYTES / 8)];
  }
  if (acc == 12345.678) buf[0] = acc;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    Rec r;
    r.hw_id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    r.xcc_id = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    r.t0 = t0;
    r.t1 = t1;
    out[blockIdx.x] = r;
  }
}

template <int LDS_BYTES, int WAVES>
void run(int blocks, int spin) {
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * blocks);
  hipLaunchKernelGGL((k_census<LDS_BYTES, WAVES>), dim3(blocks), dim3(64), 0, 0, d, spin);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_census<LDS_BYTES, WAVES>), dim3(blocks), dim3(64), 0, 0, d, spin);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Rec> h(blocks);
  hipMemcpy(h.data(), d, sizeof(Rec) * blocks, hipMemcpyDeviceToHost);
  // CU key: xcc id + se / sh / cu bits of HW_ID (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13])
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
  std::map<unsigned, int> per_simd;
  for (auto& r : h) {
    const unsigned key = ((r.xcc_id & 0xf) << 16) | (r.hw_id & 0xff00);
    ev[key].push_back({r.t0, +1});
    ev[key].push_back({r.t1, -1});
  }
  int worst = 0, best = 1 << 30;
  double mean = 0;
  for (auto& kv : ev) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
    int cur = 0, mx = 0;
    for (auto& e : v) {
      cur += e.second;
      mx = std::max(mx, cur);
    }
    worst = std::max(worst, mx);
    best = std::min(best, mx);
    mean += mx;
  }
  printf("LDS %6d B  waves_per_eu %d  blocks %5d  CUs seen %3zu  max resident per CU: min %d mean %.2f max %d   kernel %.1f us\n", LDS_BYTES, WAVES,
         blocks, ev.size(), best, mean / ev.size(), worst, ms * 1e3);
  hipFree(d);
}

int main(int argc, char** argv) {
  const int spin = argc > 1 ? atoi(argv[1]) : 20000;
  run<13120, 3>(3072, spin);
  run<13120, 3>(6144, spin);
  run<12288, 3>(3072, spin);
  run<10240, 3>(3072, spin);
  run<8192, 3>(3072, spin);
  run<4096, 3>(3072, spin);
  run<13120, 4>(4096, spin);
  run<8192, 4>(4096, spin);
  run<13824, 2>(2048, spin);
  run<13824, 2>(4096, spin);
  run<1024, 8>(8192, spin);
  return 0;
}
