// Stand-alone reproducer for the round-2 abort (gpurun_out/r2_gputest23.log): a ThreadLocal-mode stream capture on
// host thread A while host thread B makes other HIP runtime calls.  For each kind of call B makes, counts how many of
// A's captures end with an error.  No library code involved: this asks the runtime alone.
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/repro_capture tools/repro_capture_threads.hip -lpthread && /tmp/repro_capture
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void tiny_kernel(int* p, int v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v;
}
__global__ void lds_kernel(int* p) {
  extern __shared__ int sh[];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) p[1] = sh[63];
}

static const char* kNames[] = {"idle (control)",
                               "hipMalloc + hipFree",
                               "hipMalloc only (freed after the run)",
                               "kernel launches on another stream",
                               "hipStreamCreate + hipStreamDestroy",
                               "hipFuncSetAttribute",
                               "capture + instantiate + destroy of its own graph (ThreadLocal, no mutex)",
                               "hipGraphExecDestroy + hipGraphDestroy of a prebuilt graph",
                               "hipMemcpyAsync D2H + hipStreamSynchronize",
                               "hipEventCreate/Record/Synchronize/Destroy",
                               "hipDeviceSynchronize",
                               "hipMemset (synchronous, null stream)",
                               "hipMemcpy host -> device (synchronous)",
                               "hipMemcpy device -> host (synchronous)",
                               "hipStreamSynchronize(null stream)",
                               "hipMemsetAsync on the null stream",
                               "hipFree of a 64 MB block (hipMalloc before)"};
constexpr int NKIND = sizeof(kNames) / sizeof(kNames[0]);

struct Prebuilt {
  hipGraph_t g;
  hipGraphExec_t e;
};

static bool build_graph(hipStream_t cap, int* buf, int launches, hipGraph_t* g, hipError_t* first_err) {
  *first_err = hipSuccess;
  hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { *first_err = e; return false; }
  for (int i = 0; i < launches; ++i) {
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, cap, buf, 1);
    e = hipGetLastError();
    if (e != hipSuccess && *first_err == hipSuccess) *first_err = e;
  }
  e = hipStreamEndCapture(cap, g);
  if (e != hipSuccess && *first_err == hipSuccess) *first_err = e;
  return *first_err == hipSuccess;
}

int main() {
  int* bufA = nullptr;
  int* bufB = nullptr;
  hipMalloc(&bufA, 64);
  hipMalloc(&bufB, 64);
  hipMemset(bufA, 0, 64);
  hipMemset(bufB, 0, 64);
  const int captures = 40, launches = 400;
  int failures_total = 0;
  for (int kind = 0; kind < NKIND; ++kind) {
    std::atomic<bool> stop{false};
    std::atomic<long> b_ops{0};
    std::atomic<int> b_errors{0};
    // prebuilt graphs for kind 7 (built before thread A starts capturing)
    std::vector<Prebuilt> pre;
    if (kind == 7) {
      hipStream_t s;
      hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      for (int i = 0; i < 4000; ++i) {
        Prebuilt p{};
        hipError_t fe;
        if (!build_graph(s, bufB, 8, &p.g, &fe)) break;
        if (hipGraphInstantiate(&p.e, p.g, nullptr, nullptr, 0) != hipSuccess) break;
        pre.push_back(p);
      }
      hipStreamDestroy(s);
    }
    std::thread tb([&] {
      hipSetDevice(0);
      hipStream_t s = nullptr;
      hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      std::vector<void*> keep;
      size_t next_pre = 0;
      int host = 0;
      while (!stop.load()) {
        hipError_t e = hipSuccess;
        switch (kind) {
          case 0: std::this_thread::yield(); break;
          case 1: { void* p = nullptr; e = hipMalloc(&p, 1 << 20); if (e == hipSuccess) e = hipFree(p); break; }
          case 2: { void* p = nullptr; if (keep.size() < 20000) { e = hipMalloc(&p, 1 << 16); keep.push_back(p); } break; }
          case 3: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, bufB, 1); e = hipGetLastError(); break;
          case 4: { hipStream_t t; e = hipStreamCreateWithFlags(&t, hipStreamNonBlocking); if (e == hipSuccess) e = hipStreamDestroy(t); break; }
          case 5: e = hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 + 64 * (int)(b_ops.load() & 15)); break;
          case 6: {
            hipGraph_t g; hipGraphExec_t x; hipError_t fe;
            if (build_graph(s, bufB, 50, &g, &fe)) {
              e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
              if (e == hipSuccess) { hipGraphExecDestroy(x); }
              hipGraphDestroy(g);
            } else e = fe;
            break;
          }
          case 7:
            if (next_pre < pre.size()) { e = hipGraphExecDestroy(pre[next_pre].e); hipGraphDestroy(pre[next_pre].g); ++next_pre; }
            break;
          case 8: e = hipMemcpyAsync(&host, bufB, 4, hipMemcpyDeviceToHost, s); if (e == hipSuccess) e = hipStreamSynchronize(s); break;
          case 9: { hipEvent_t ev; e = hipEventCreate(&ev); if (e == hipSuccess) { hipEventRecord(ev, s); hipEventSynchronize(ev); hipEventDestroy(ev); } break; }
          case 10: e = hipDeviceSynchronize(); break;
          case 11: e = hipMemset(bufB + 8, 0, 8); break;
          case 12: e = hipMemcpy(bufB + 8, &host, 4, hipMemcpyHostToDevice); break;
          case 13: e = hipMemcpy(&host, bufB + 8, 4, hipMemcpyDeviceToHost); break;
          case 14: e = hipStreamSynchronize(nullptr); break;
          case 15: e = hipMemsetAsync(bufB + 8, 0, 8, nullptr); break;
          case 16: { void* p = nullptr; e = hipMalloc(&p, 64u << 20); if (e == hipSuccess) e = hipFree(p); break; }
        }
        if (e != hipSuccess) b_errors.fetch_add(1);
        b_ops.fetch_add(1);
      }
      hipStreamSynchronize(s);
      for (void* p : keep) hipFree(p);
      hipStreamDestroy(s);
    });
    int fails = 0;
    hipError_t first = hipSuccess;
    {
      hipStream_t cap;
      hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
      hipStream_t run;
      hipStreamCreateWithFlags(&run, hipStreamNonBlocking);
      for (int c = 0; c < captures; ++c) {
        hipGraph_t g = nullptr;
        hipGraphExec_t x = nullptr;
        hipError_t fe;
        bool ok = build_graph(cap, bufA, launches, &g, &fe);
        if (ok) {
          hipError_t e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
          if (e == hipSuccess) e = hipGraphLaunch(x, run);
          if (e == hipSuccess) e = hipStreamSynchronize(run);
          if (e != hipSuccess) { ok = false; fe = e; }
        }
        if (!ok) {
          ++fails;
          if (first == hipSuccess) first = fe;
          (void)hipGetLastError();
        }
        if (x) hipGraphExecDestroy(x);
        if (g) hipGraphDestroy(g);
      }
      hipStreamDestroy(cap);
      hipStreamDestroy(run);
    }
    stop.store(true);
    tb.join();
    printf("B: %-78s A: %2d / %d captures failed%s%s   (B made %ld calls, %d of them returned an error)\n", kNames[kind],
           fails, captures, fails ? ", first error: " : "", fails ? hipGetErrorString(first) : "", b_ops.load(),
           b_errors.load());
    fflush(stdout);
    failures_total += fails;
  }
  hipFree(bufA);
  hipFree(bufB);
  printf("total failed captures: %d\n", failures_total);
  return 0;
}
