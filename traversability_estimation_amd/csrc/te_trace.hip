// te_trace.hip -- roctx ranges around the phases of a launch (tracing row: the reference times its
// chain with ros::WallTime and prints through ROS_DEBUG, TE/src/TraversabilityMap.cpp:205-235; here the phases show up in
// `rocprofv3 --marker-trace`).  The marker library is looked up at run time -- librocprofiler-sdk-roctx (what rocprofv3
// listens to), else the older libroctx64 -- so libtravgpu.so has no link-time dependency on a profiler; without either
// library a range is two predictable branches.
#include <dlfcn.h>

#include <mutex>

#include "te_internal.h"

namespace te {
namespace {
typedef int (*push_fn)(const char*);
typedef int (*pop_fn)(void);
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
std::once_flag g_once;

void resolve() {
  for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
    void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    push_fn a = (push_fn)dlsym(h, "roctxRangePushA");
    pop_fn b = (pop_fn)dlsym(h, "roctxRangePop");
    if (a && b) {
      g_push = a;
      g_pop = b;
      return;
    }
  }
}
}  // namespace

TraceRange::TraceRange(const char* name) : live_(false) {
  std::call_once(g_once, resolve);
  if (g_push) {
    (void)g_push(name);
    live_ = true;
  }
}

TraceRange::~TraceRange() {
  if (live_) (void)g_pop();
}

bool trace_available() {
  std::call_once(g_once, resolve);
  return g_push != nullptr;
}

}  // namespace te
