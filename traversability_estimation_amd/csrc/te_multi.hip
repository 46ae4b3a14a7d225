// te_multi.hip -- C-ABI of libtravgpu.so, the batch axis over several contexts: maps are independent units,
// so a batch is cut into contiguous blocks (te_shard_range), rank 0's parameter block is broadcast once (te_bcast_params:
// RCCL, looked up at run time) and every context runs its own shard (te_run_chain_multi / te_sync_multi) -- no data-path
// collective.  The reference has no parallelism at all (TraversabilityMap.cpp:214 is one single-threaded call).
#include "te_ctx.h"

using namespace te;
using namespace te::shim;

extern "C" {

int te_shard_range(int batch, int n_shards, int k, int* first, int* count) {
  if (!first || !count || batch < 0 || n_shards <= 0 || k < 0 || k >= n_shards)
    return fail(TE_ERR_INVALID_ARG, "te_shard_range: batch=%d n_shards=%d k=%d", batch, n_shards, k);
  const int base = batch / n_shards, extra = batch % n_shards;
  *first = k * base + (k < extra ? k : extra);
  *count = base + (k < extra ? 1 : 0);
  return TE_OK;
}

namespace {
// the few RCCL entry points the parameter broadcast needs, resolved at run time (no link-time dependency: a single-GPU
// host never loads the library)
struct Rccl {
  typedef void* comm_t;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  bool ok = false;
  Rccl() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    CommInitAll = (int (*)(comm_t*, int, const int*))dlsym(h, "ncclCommInitAll");
    CommDestroy = (int (*)(comm_t))dlsym(h, "ncclCommDestroy");
    GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    Broadcast = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(h, "ncclBroadcast");
    ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
  }
};
}  // namespace

int te_bcast_params(te_ctx** ctxs, int n, int root) {
  if (!ctxs || n <= 0 || root < 0 || root >= n) return fail(TE_ERR_INVALID_ARG, "te_bcast_params: n=%d root=%d", n, root);
  for (int k = 0; k < n; ++k)
    if (!ctxs[k]) return fail(TE_ERR_INVALID_ARG, "te_bcast_params: NULL context %d", k);
  te_params p;
  int rc = te_get_params(ctxs[root], &p);
  if (rc) return rc;
  // one representative context per device (the root for its own); the others on a device are served from the host copy
  std::vector<int> devs, rep;
  devs.push_back(ctxs[root]->device);
  rep.push_back(root);
  for (int k = 0; k < n; ++k) {
    bool seen = false;
    for (int d : devs) seen = seen || d == ctxs[k]->device;
    if (!seen) {
      devs.push_back(ctxs[k]->device);
      rep.push_back(k);
    }
  }
  // (the receivers' copies start out zeroed: what a context is given below has been through the broadcast)
  std::vector<te_params> got(devs.size());
  for (auto& q : got) memset(&q, 0, sizeof(q));
  got[0] = p;
  // TE_OPT_BCAST_RCCL on the root: a communicator of ONE rank when every context shares the root's device -- the same
  // calls, so that a one-GPU host exercises this branch (the root's device then receives its own block back)
  if (devs.size() > 1 || ctxs[root]->opt_bcast_rccl) {
    static Rccl rccl;
    if (!rccl.ok) return fail(TE_ERR_UNSUPPORTED, "te_bcast_params: %zu devices but librccl could not be loaded", devs.size());
    const int nd = (int)devs.size();
    std::vector<Rccl::comm_t> comms(nd, nullptr);
    std::vector<void*> buf(nd, nullptr);
    // streams of this call's own: the contexts' streams belong to their mutexes, and this is configure-time
    std::vector<hipStream_t> st(nd, nullptr);
    int e = rccl.CommInitAll(comms.data(), nd, devs.data());
    if (e) return fail(TE_ERR_HIP, "te_bcast_params: ncclCommInitAll failed (%d)", e);
    bool bad = false;
    for (int d = 0; d < nd && !bad; ++d) {
      bad = hipSetDevice(devs[d]) != hipSuccess || hipMalloc(&buf[d], sizeof(te_params)) != hipSuccess ||
            hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking) != hipSuccess;
      if (!bad && d == 0) bad = hipMemcpy(buf[0], &p, sizeof(te_params), hipMemcpyHostToDevice) != hipSuccess;
    }
    if (!bad) {
      rccl.GroupStart();
      for (int d = 0; d < nd; ++d) {  // rank 0 is the root's device; ncclChar == 0
        (void)hipSetDevice(devs[d]);
        e = e ? e : rccl.Broadcast(buf[d], buf[d], sizeof(te_params), 0, 0, comms[d], st[d]);
      }
      e = rccl.GroupEnd() || e;
      for (int d = 0; d < nd && !e; ++d) {
        (void)hipSetDevice(devs[d]);
        bad = bad || hipStreamSynchronize(st[d]) != hipSuccess ||
              hipMemcpy(&got[d], buf[d], sizeof(te_params), hipMemcpyDeviceToHost) != hipSuccess;
      }
    }
    for (int d = 0; d < nd; ++d) {
      (void)hipSetDevice(devs[d]);
      if (buf[d]) (void)hipFree(buf[d]);
      if (st[d]) (void)hipStreamDestroy(st[d]);
      if (comms[d]) rccl.CommDestroy(comms[d]);
    }
    if (bad || e) {
      (void)hipGetLastError();
      return fail(TE_ERR_HIP, "te_bcast_params: RCCL broadcast failed (%d)", e);
    }
  }
  for (int k = 0; k < n; ++k) {
    if (k == root) continue;
    size_t d = 0;
    while (d < devs.size() && devs[d] != ctxs[k]->device) ++d;
    rc = te_set_params(ctxs[k], &got[d]);
    if (rc) return rc;
  }
  return TE_OK;
}

int te_run_chain_multi(te_ctx** ctxs, int n, unsigned flags) {
  if (!ctxs || n <= 0) return fail(TE_ERR_INVALID_ARG, "te_run_chain_multi: n=%d", n);
  for (int k = 0; k < n; ++k) {
    const int rc = te_run_chain(ctxs[k], flags);
    if (rc) return rc;
  }
  return TE_OK;
}

int te_sync_multi(te_ctx** ctxs, int n) {
  if (!ctxs || n <= 0) return fail(TE_ERR_INVALID_ARG, "te_sync_multi: n=%d", n);
  for (int k = 0; k < n; ++k) {
    const int rc = te_sync(ctxs[k]);
    if (rc) return rc;
  }
  return TE_OK;
}

}  // extern "C"
