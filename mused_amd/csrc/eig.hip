// Batched symmetric eigensolver: cyclic two-sided Jacobi in fp64, round-robin (chess
// tournament) ordering, ONE KERNEL PER ROTATION SET.
//
// Used for every small dense eigenproblem on the path: the 2l x 2l Gram matrix of an FD
// rotation (the "per-window SVD/rotation step" of SWFD), the <=4l x 4l Gram of a sketch
// query and the (l+10) x (l+10) Gram of the randomized-SVD projection B = Q^T A.
//
// Why this shape: a 256 x 256 fp64 matrix (512 KB) does not fit one CU's LDS (160 KB), and
// Jacobi needs a grid-wide dependency between rotation sets.  On gfx950 a same-stream kernel
// boundary (~1.5 us, replayed from a hipGraph) is cheaper than any in-kernel grid barrier,
// so each set of n/2 disjoint rotations is one launch over ALL matrices of the batch:
// thread (P, Q) owns the 2 x 2 block {p,q} x {p',q'} of the current pairing, recomputes the
// two rotations it needs from the three diagonal-block entries of each pair, and writes
// G' = J^T G J and V' = V J into the other half of a ping-pong buffer (no in-place hazard).
// Only blocks P <= Q are computed; the mirror image is stored, so G stays exactly symmetric.
#include <vector>

#include "internal.h"

namespace mused {

struct EigPlan {
  int n, batch, sweeps;
  double* G[2];
  double* V[2];
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStream_t cap_stream;
  bool have_graph;
};

__device__ __forceinline__ void rr_pair(int n, int step, int k, int& p, int& q) {
  // round-robin tournament on n (even) players, n-1 steps, pair k of step `step`
  const int m = n - 1;
  if (k == 0) {
    p = m;
    q = step % m;
  } else {
    p = (step + k) % m;
    q = (step - k + m) % m;
  }
  if (p > q) { const int t = p; p = q; q = t; }
}

__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
  // rotation zeroing a_pq:  J = [[c, s], [-s, c]] applied as J^T A J
  if (apq == 0.0 || fabs(apq) <= 1e-300) {
    c = 1.0; s = 0.0;
    return;
  }
  const double tau = (aqq - app) / (2.0 * apq);
  const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  c = 1.0 / sqrt(1.0 + t * t);
  s = t * c;
}

__global__ __launch_bounds__(256) void jacobi_step_kernel(const double* __restrict__ Gin, double* __restrict__ Gout,
                                                         const double* __restrict__ Vin, double* __restrict__ Vout,
                                                         int n, int step) {
  const int h = n >> 1;
  const int P = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int Q = blockIdx.x * 16 + (threadIdx.x & 15);
  if (P >= h || Q >= h) return;
  const long off = (long)blockIdx.z * n * n;
  const double* G = Gin + off;
  double* Go = Gout + off;
  const double* V = Vin + off;
  double* Vo = Vout + off;

  int p, q, p2, q2;
  rr_pair(n, step, P, p, q);
  rr_pair(n, step, Q, p2, q2);
  double cq, sq;
  jacobi_cs(G[(long)p2 * n + p2], G[(long)q2 * n + q2], G[(long)p2 * n + q2], cq, sq);

  // V' = V J_Q on rows 2P, 2P+1 (any row partition works; rows are independent)
  {
    const int r0 = 2 * P;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const long ro = (long)(r0 + rr) * n;
      const double vp = V[ro + p2], vq = V[ro + q2];
      Vo[ro + p2] = cq * vp - sq * vq;
      Vo[ro + q2] = sq * vp + cq * vq;
    }
  }
  if (P > Q) return;
  if (P == Q) {
    const double app = G[(long)p * n + p], aqq = G[(long)q * n + q], apq = G[(long)p * n + q];
    if (sq == 0.0 && cq == 1.0) {
      Go[(long)p * n + p] = app; Go[(long)q * n + q] = aqq;
      Go[(long)p * n + q] = apq; Go[(long)q * n + p] = apq;
    } else {
      const double t = sq / cq;
      Go[(long)p * n + p] = app - t * apq;
      Go[(long)q * n + q] = aqq + t * apq;
      Go[(long)p * n + q] = 0.0;
      Go[(long)q * n + p] = 0.0;
    }
    return;
  }
  double cp, sp;
  jacobi_cs(G[(long)p * n + p], G[(long)q * n + q], G[(long)p * n + q], cp, sp);
  // B = G[{p,q}, {p2,q2}];  B' = R_P^T B R_Q with R = [[c, s], [-s, c]]
  const double b00 = G[(long)p * n + p2], b01 = G[(long)p * n + q2];
  const double b10 = G[(long)q * n + p2], b11 = G[(long)q * n + q2];
  // T = B R_Q
  const double t00 = cq * b00 - sq * b01, t01 = sq * b00 + cq * b01;
  const double t10 = cq * b10 - sq * b11, t11 = sq * b10 + cq * b11;
  // B' = R_P^T T
  const double n00 = cp * t00 - sp * t10, n01 = cp * t01 - sp * t11;
  const double n10 = sp * t00 + cp * t10, n11 = sp * t01 + cp * t11;
  Go[(long)p * n + p2] = n00; Go[(long)p2 * n + p] = n00;
  Go[(long)p * n + q2] = n01; Go[(long)q2 * n + p] = n01;
  Go[(long)q * n + p2] = n10; Go[(long)p2 * n + q] = n10;
  Go[(long)q * n + q2] = n11; Go[(long)q2 * n + q] = n11;
}

__global__ void eig_init_kernel(double* __restrict__ V, int n, long total) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const long e = gid % ((long)n * n);
  V[gid] = ((e / n) == (e % n)) ? 1.0 : 0.0;
}

__global__ void eig_extract_kernel(const double* __restrict__ G, const double* __restrict__ V, int n, int batch,
                                   double* __restrict__ evals, double* __restrict__ Vout) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)batch * n * n;
  if (gid >= total) return;
  if (Vout) Vout[gid] = V[gid];
  const long e = gid % ((long)n * n);
  const int r = (int)(e / n), c = (int)(e % n);
  if (r == c) evals[(gid / ((long)n * n)) * n + r] = G[gid];
}

static int enqueue_sweeps(EigPlan* p, hipStream_t st) {
  const int n = p->n, h = n / 2;
  dim3 grid(cdiv(h, 16), cdiv(h, 16), p->batch);
  int cur = 0;
  for (int sw = 0; sw < p->sweeps; ++sw) {
    for (int step = 0; step < n - 1; ++step) {
      hipLaunchKernelGGL(jacobi_step_kernel, grid, dim3(256), 0, st, p->G[cur], p->G[cur ^ 1], p->V[cur],
                         p->V[cur ^ 1], n, step);
      cur ^= 1;
    }
  }
  MUSED_LAUNCH_CHECK();
  return cur;
}

int eig_plan_create(int n, int batch, int sweeps, bool own_graph, EigPlan** out) {
  MUSED_REQUIRE(n >= 2 && n % 2 == 0 && batch >= 1 && sweeps >= 1, "eig_plan_create: n must be even (n=%d)", n);
  EigPlan* p = new EigPlan();
  memset(p, 0, sizeof(*p));
  p->n = n; p->batch = batch; p->sweeps = sweeps;
  const size_t bytes = sizeof(double) * (size_t)batch * n * n;
  for (int i = 0; i < 2; ++i) {
    MUSED_CHECK_HIP(hipMalloc(&p->G[i], bytes));
    MUSED_CHECK_HIP(hipMalloc(&p->V[i], bytes));
  }
  p->have_graph = false;
  const char* ng = getenv("MUSED_NO_GRAPH");
  if (own_graph && !(ng && ng[0] == '1')) {
    MUSED_CHECK_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
    MUSED_CHECK_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_sweeps(p, p->cap_stream);
    hipError_t e = hipStreamEndCapture(p->cap_stream, &p->graph);
    if (rc < 0 || e != hipSuccess) {
      set_error("eig_plan_create: graph capture failed (%s)", hipGetErrorString(e));
      return MUSED_ERR_HIP;
    }
    MUSED_CHECK_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    p->have_graph = true;
  }
  *out = p;
  return MUSED_OK;
}

void eig_plan_destroy(EigPlan* p) {
  if (!p) return;
  if (p->have_graph) {
    (void)hipGraphExecDestroy(p->exec);
    (void)hipGraphDestroy(p->graph);
    (void)hipStreamDestroy(p->cap_stream);
  }
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(p->G[i]);
    (void)hipFree(p->V[i]);
  }
  delete p;
}

double* eig_plan_input(EigPlan* p) { return p->G[0]; }

int eig_plan_run_inplace(EigPlan* p, double* evals, double* V, hipStream_t st, bool allow_graph) {
  const long total = (long)p->batch * p->n * p->n;
  hipLaunchKernelGGL(eig_init_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p->V[0], p->n, total);
  int fin;
  if (p->have_graph && allow_graph) {
    MUSED_CHECK_HIP(hipGraphLaunch(p->exec, st));
    fin = (p->sweeps * (p->n - 1)) & 1;
  } else {
    fin = enqueue_sweeps(p, st);
    if (fin < 0) return fin;
  }
  hipLaunchKernelGGL(eig_extract_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p->G[fin], p->V[fin], p->n,
                     p->batch, evals, V);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int eig_plan_run(EigPlan* p, const double* G, double* evals, double* V, hipStream_t st) {
  const size_t bytes = sizeof(double) * (size_t)p->batch * p->n * p->n;
  MUSED_CHECK_HIP(hipMemcpyAsync(p->G[0], G, bytes, hipMemcpyDeviceToDevice, st));
  return eig_plan_run_inplace(p, evals, V, st, true);
}

}  // namespace mused

using namespace mused;

extern "C" {

// Unit-testable primitive: eigen-decomposition of `batch` symmetric n x n fp64 matrices
// (n even).  evals: batch x n (unsorted), V: batch x n x n with V[:, j] the eigenvector of
// evals[j].  Creates and destroys a plan per call (test/diagnostic use only).
int mused_syevj_batched(const double* G, int n, int batch, int sweeps, double* evals, double* V, void* stream) {
  MUSED_REQUIRE(G && evals && V, "mused_syevj_batched: null pointer");
  EigPlan* p = nullptr;
  int rc = eig_plan_create(n, batch, sweeps, true, &p);
  if (rc) return rc;
  rc = eig_plan_run(p, G, evals, V, (hipStream_t)stream);
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  eig_plan_destroy(p);
  if (rc) return rc;
  MUSED_CHECK_HIP(e);
  return MUSED_OK;
}

}  // extern "C"
