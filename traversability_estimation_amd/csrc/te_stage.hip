// te_stage.hip -- host <-> device transfers of whole layers through PAGEABLE caller buffers.
//
// The reference's plugins hand over grid_map::Matrix buffers (Eigen heap memory, pageable):
//   mapOut = mapIn; mapOut.add(type_)     traversability_estimation_filters/src/SlopeFilter.cpp:62-63 (every plugin)
//   TraversabilityMap::setElevationMap     traversability_estimation/src/TraversabilityMap.cpp:135-154
// hipMemcpyAsync from / to pageable memory goes through the runtime's own staging, one bounce at a time: 64 MB up and
// 5 x 64 MB down took 51 ms per 4096^2 frame in round 3 (7.5 GB/s), 131 x the resident launch.  Here the shim keeps a ring
// of page-locked slots per context and a small pool of copy threads per process: chunk k is copied host -> slot by the
// pool while chunk k - 1 crosses PCIe from its slot (uploads), or the DMA of chunk k + kSlots - 1 runs while the pool
// copies chunk k slot -> host (downloads).  Buffers the caller has page-locked (te_pin_host) skip all of this.
#include <pthread.h>
#include <sched.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "te_internal.h"

namespace te {

namespace {

// memcpy split over a few persistent threads (one job at a time: the contexts' transfers queue up on the mutex).
// A transfer is a train of 8 MiB jobs a fraction of a millisecond apart, so a worker that has finished its piece SPINS on
// the job counter for a while before it goes back to sleep: waking a dozen threads through a condition variable costs
// more than the 0.1 ms their pieces take (measured: 64 MB up in 2.1 ms with sleeping workers, one thread alone copies at
// 24 GB/s on these hosts).
class CopyPool {
 public:
  // Two pools per process: 0 for the transfers a call waits for, 1 for the prefetches that run beside them
  // (te_prefetch_layers: PCIe is full duplex, one pool's copies are not)
  static CopyPool& get(int which = 0) {
    static CopyPool* p[2] = {new CopyPool, nullptr};  // (never destroyed: their threads sleep on a condition variable until the process ends)
    if (which == 0) return *p[0];
    static std::once_flag once;
    std::call_once(once, [] { p[1] = new CopyPool; });
    return *p[1];
  }
  // fork(): the child has the pool's state but none of its threads -- it copies on the calling thread from then on
  // (registered once, by the constructor; the locks are not taken around the fork: a child only ever reads `forked_`)
  static void after_fork_in_child() { forked_.store(true, std::memory_order_relaxed); }
  void copy(void* dst, const void* src, size_t bytes) {
    if (bytes < (1u << 20) || n_workers_ == 0 || forked_.load(std::memory_order_relaxed)) {
      memcpy(dst, src, bytes);
      return;
    }
    std::lock_guard<std::mutex> job(job_mu_);
    const size_t parts = (size_t)n_workers_ + 1;
    const size_t piece = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
    dst_ = (char*)dst;
    src_ = (const char*)src;
    bytes_ = bytes;
    piece_ = piece;
    pending_.store(n_workers_);
    generation_.fetch_add(1);  // (sequentially consistent, like the sleepers' counter: one side always sees the other)
    if (sleepers_.load() != 0) {
      std::lock_guard<std::mutex> lk(mu_);  // (a worker between its last look at the counter and its wait holds mu_)
      cv_.notify_all();
    }
    part(parts - 1);  // the caller takes the last piece
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  CopyPool() {
    // the CPUs this process may run on (a cpuset / taskset narrower than the machine), not the machine's
    unsigned hw = std::thread::hardware_concurrency();
    {
      cpu_set_t set;
      CPU_ZERO(&set);
      if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0 && (unsigned)c < hw) hw = (unsigned)c;
      }
    }
    // (a copy thread moves 20-25 GB/s from cache-cold memory on these hosts; PCIe takes 50: a few of them keep the DMA
    // engine fed.  More threads than the container may really use do harm -- the GPU boxes of this pool show 256 cores
    // and run OpenMP fastest on 32 -- hence the modest numbers)
    const int n = hw >= 32 ? 7 : (hw >= 8 ? 3 : (hw >= 4 ? 1 : 0));
    // A thread that cannot be created (EAGAIN: the process' thread limit) must not take the process down through the
    // extern "C" boundary: the pool keeps the workers it got -- none at all means a plain memcpy on the caller's thread.
    for (int k = 0; k < n; ++k) {
      try {
        std::thread t([this, k] { run(k); });
        t.detach();  // (process-lifetime pool: the library may be unloaded at exit while they sleep)
        ++n_workers_;
      } catch (...) {
        break;
      }
    }
    (void)pthread_atfork(nullptr, nullptr, &CopyPool::after_fork_in_child);
  }
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#else
    std::this_thread::yield();
#endif
  }
  void part(size_t k) {
    const size_t off = k * piece_;
    if (off < bytes_) memcpy(dst_ + off, src_ + off, bytes_ - off < piece_ ? bytes_ - off : piece_);
  }
  void run(int k) {
    unsigned long long seen = 0;
    for (;;) {
      // spin for about a millisecond (the gap between two jobs of one transfer is a DMA chunk: 0.15 ms), then sleep
      bool got = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spin = 0;; ++spin) {
        if (generation_.load(std::memory_order_acquire) != seen) {
          got = true;
          break;
        }
        if (spin < 4096)
          cpu_relax();
        else
          std::this_thread::yield();  // (a host with fewer free cores than workers: let the copying threads run)
        if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return generation_.load() != seen; });
        sleepers_.fetch_sub(1);
      }
      seen = generation_.load(std::memory_order_acquire);
      part((size_t)k);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  int n_workers_ = 0;  // workers k = 0 .. n_workers_ - 1 were created (in order: a failed creation ends the loop)
  static std::atomic<bool> forked_;
  std::mutex job_mu_, mu_;
  std::condition_variable cv_;
  char* dst_ = nullptr;
  const char* src_ = nullptr;
  size_t bytes_ = 0, piece_ = 0;
  std::atomic<int> pending_{0}, sleepers_{0};
  std::atomic<unsigned long long> generation_{0};
};

std::atomic<bool> CopyPool::forked_{false};

bool byte_is_pinned(const void* p) {
  hipPointerAttribute_t at;
  const hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer: "invalid value")
    return false;
  }
  return at.type == hipMemoryTypeHost;
}
// the WHOLE buffer lies in page-locked memory (first and last byte: a buffer that only starts inside a registered range
// takes the staged path)
bool host_is_pinned(const void* p, size_t bytes) {
  return byte_is_pinned(p) && (bytes == 0 || byte_is_pinned((const char*)p + bytes - 1));
}

}  // namespace

HostStager::~HostStager() { release(); }

void HostStager::release() {
  for (int s = 0; s < kSlots; ++s) {
    if (slot[s]) (void)hipHostFree(slot[s]);
    if (ev[s]) (void)hipEventDestroy(ev[s]);
    slot[s] = nullptr;
    ev[s] = nullptr;
  }
  if (ev_order) (void)hipEventDestroy(ev_order);
  ev_order = nullptr;
  if (stream) (void)hipStreamDestroy(stream);
  stream = nullptr;
}

hipError_t HostStager::ensure() {
  if (stream) return hipSuccess;
  hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_order, hipEventDisableTiming);
  for (int s = 0; s < kSlots && e == hipSuccess; ++s) {
    e = hipHostMalloc((void**)&slot[s], kChunk, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[s], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    release();
  }
  return e;
}

// host -> device, ordered after everything queued on `compute` so far; `compute` waits for the last chunk
hipError_t HostStager::upload(void* dev, const void* host, size_t bytes, hipStream_t compute) {
  if (bytes < kMinBytes || host_is_pinned(host, bytes) || ensure() != hipSuccess) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, compute);
  hipError_t e = hipEventRecord(ev_order, compute);  // (kernels queued earlier may still read the layer)
  if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_order, 0);
  const size_t nchunks = (bytes + kChunk - 1) / kChunk;
  for (size_t k = 0; k < nchunks && e == hipSuccess; ++k) {
    const int s = (int)(k % kSlots);
    const size_t off = k * kChunk, n = bytes - off < kChunk ? bytes - off : kChunk;
    if (k >= (size_t)kSlots) e = hipEventSynchronize(ev[s]);  // the slot's previous chunk has left it
    if (e != hipSuccess) break;
    CopyPool::get(pool).copy(slot[s], (const char*)host + off, n);
    e = hipMemcpyAsync((char*)dev + off, slot[s], n, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(ev[s], stream);
  }
  // `compute` waits for whatever has been queued on the copy stream -- also after an error: kernels queued later must not
  // overtake chunks that are still in flight
  {
    hipError_t e2 = hipEventRecord(ev_order, stream);
    if (e2 == hipSuccess) e2 = hipStreamWaitEvent(compute, ev_order, 0);
    if (e == hipSuccess) e = e2;
  }
  // the slots are reused by the next transfer, and the caller may reuse `host` as soon as we return: both are safe --
  // every chunk has been copied out of `host`, and a slot is only rewritten after its event.  (On an error the drain
  // matters even more: the next transfer starts from idle slots and events.)
  {
    const hipError_t e2 = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = e2;
  }
  return e;
}

// device -> host, ordered after everything queued on `compute` so far; returns when `host` holds the data
hipError_t HostStager::download(void* host, const void* dev, size_t bytes, hipStream_t compute) {
  if (bytes < kMinBytes || host_is_pinned(host, bytes) || ensure() != hipSuccess) {
    hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, compute);
    return e == hipSuccess ? hipStreamSynchronize(compute) : e;
  }
  hipError_t e = hipEventRecord(ev_order, compute);
  if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_order, 0);
  const size_t nchunks = (bytes + kChunk - 1) / kChunk;
  auto issue = [&](size_t k) {
    const int s = (int)(k % kSlots);
    const size_t off = k * kChunk, n = bytes - off < kChunk ? bytes - off : kChunk;
    hipError_t r = hipMemcpyAsync(slot[s], (const char*)dev + off, n, hipMemcpyDeviceToHost, stream);
    return r == hipSuccess ? hipEventRecord(ev[s], stream) : r;
  };
  for (size_t k = 0; k < nchunks && k < (size_t)kSlots && e == hipSuccess; ++k) e = issue(k);
  for (size_t k = 0; k < nchunks && e == hipSuccess; ++k) {
    const int s = (int)(k % kSlots);
    const size_t off = k * kChunk, n = bytes - off < kChunk ? bytes - off : kChunk;
    e = hipEventSynchronize(ev[s]);
    if (e != hipSuccess) break;
    CopyPool::get(pool).copy((char*)host + off, slot[s], n);
    if (k + kSlots < nchunks) e = issue(k + kSlots);
  }
  if (e != hipSuccess) (void)hipStreamSynchronize(stream);  // chunks still in flight into the slots: drain before the next transfer reuses them
  return e;
}

}  // namespace te
