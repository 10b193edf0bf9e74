// Stand-alone check of the round-1 claim "hipMemsetAsync nodes captured into a hipGraph wrote to stale addresses on
// replay" (DESIGN.md section 5).  Captures {memset of a small int buffer; kernel that raises a flag in it} between two
// guard allocations, replays the graph many times while other allocations come and go, and verifies (a) the memset
// node's destination (hipGraphMemsetNodeGetParams) is the application pointer, (b) the guards are never touched,
// (c) the buffer is zero before the kernel of every replay.
//   hipcc --offload-arch=gfx950 -O2 tools/repro_graph_memset.hip -o /tmp/repro && /tmp/repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void touch(int* flags, int* seen_nonzero) {
  if (threadIdx.x < 4) {
    if (flags[threadIdx.x] != 0) atomicAdd(seen_nonzero, 1);  // the memset before us must have cleared it
    flags[threadIdx.x] = 1 + threadIdx.x;
  }
}

int main() {
  const size_t guard_ints = 1 << 16;
  int *g0, *flags, *g1, *seen;
  CK(hipMalloc(&g0, guard_ints * 4)); CK(hipMalloc(&flags, 16)); CK(hipMalloc(&g1, guard_ints * 4)); CK(hipMalloc(&seen, 4));
  CK(hipMemset(g0, 0x5a, guard_ints * 4)); CK(hipMemset(g1, 0x5a, guard_ints * 4)); CK(hipMemset(seen, 0, 4));
  CK(hipMemset(flags, 0xff, 16));
  hipStream_t cap, run;
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(flags, 0, 16, cap));
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, cap, flags, seen);
  CK(hipStreamEndCapture(cap, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipStreamDestroy(cap));  // as the library does: the capture stream dies, the graph lives on
  size_t nn = 0;
  CK(hipGraphGetNodes(graph, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(graph, nodes.data(), &nn));
  for (size_t i = 0; i < nn; ++i) {
    hipGraphNodeType t;
    CK(hipGraphNodeGetType(nodes[i], &t));
    if (t == hipGraphNodeTypeMemset) {
      hipMemsetParams mp;
      CK(hipGraphMemsetNodeGetParams(nodes[i], &mp));
      printf("memset node: dst %p (application pointer %p) value %u elementSize %u width %zu height %zu -> %s\n", mp.dst,
             (void*)flags, mp.value, mp.elementSize, mp.width, mp.height, mp.dst == (void*)flags ? "same" : "DIFFERENT");
    }
  }
  std::vector<void*> churn;
  for (int it = 0; it < 2000; ++it) {
    if (it % 7 == 0) { void* p; CK(hipMalloc(&p, 4096 + 256 * (it % 13))); churn.push_back(p); }
    if (it % 11 == 0 && !churn.empty()) { CK(hipFree(churn.back())); churn.pop_back(); }
    CK(hipGraphLaunch(exec, run));
  }
  CK(hipStreamSynchronize(run));
  std::vector<int> h(guard_ints);
  int bad = 0, seen_h = 0, fl[4];
  CK(hipMemcpy(h.data(), g0, guard_ints * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < guard_ints; ++i) bad += (h[i] != 0x5a5a5a5a);
  CK(hipMemcpy(h.data(), g1, guard_ints * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < guard_ints; ++i) bad += (h[i] != 0x5a5a5a5a);
  CK(hipMemcpy(&seen_h, seen, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(fl, flags, 16, hipMemcpyDeviceToHost));
  int rt = 0, drv = 0;
  (void)hipRuntimeGetVersion(&rt); (void)hipDriverGetVersion(&drv);
  printf("HIP runtime %d driver %d: 2000 replays, guard words changed: %d, replays that saw a non-zero flag word: %d, final flags %d %d %d %d\n",
         rt, drv, bad, seen_h, fl[0], fl[1], fl[2], fl[3]);
  printf(bad == 0 && seen_h == 0 ? "NOT REPRODUCED: captured memset nodes behaved correctly here\n" : "REPRODUCED\n");
  return 0;
}
