// the assign / update iterations of perform_clustering (matrix_operations.py:149-153 =
// sklearn KMeans(n_clusters, random_state=seed): k-means++ seeding, n_init = 1, Lloyd, max_iter = 300, tol = 1e-4) on
// the device.  The SEEDING stays on the host -- sklearn's own kmeans_plusplus driven by the same RandomState, on the
// centred embedding -- because it is what couples the result to NumPy's MT19937 stream; what runs here is
// sklearn:cluster/_kmeans.py `_kmeans_single_lloyd` (:586-720) / `_k_means_lloyd.pyx` `lloyd_iter_chunked_dense`:
//
//   repeat (<= max_iter):
//     E step   labels_i = argmin_j ( |c_j|^2 - 2 x_i . c_j ),  first minimum on ties        (_k_means_lloyd.pyx _update_chunk_dense)
//     M step   c_j <- mean of the rows labelled j;  shift = sum_j |c_j_new - c_j|^2
//     stop     if labels == labels of the previous iteration (strict convergence)  else if shift <= tol
//   if not strictly converged: one more E step with the final centres                     (_kmeans.py:709-720)
//
// Floating point: fp64 throughout like sklearn; sums are taken in a FIXED order (rows of a 256-row chunk in sequence,
// chunks in sequence), so results are reproducible run to run -- sklearn's own sums depend on its thread count, and
// labels are only sensitive to that for rows within rounding of a cell boundary (tests: bit-identical labels on every
// golden window and on the 20-window benchmark stream).  An empty cluster (sklearn relocates its centre,
// _k_means_common.pyx `_relocate_empty_clusters_dense`) is not handled here: it raises info[2] and the host falls back to
// scikit-learn for that window.
#include <stdlib.h>

#include <mutex>

#include "internal.h"

namespace mused {

constexpr int KM_CHUNK = 256;

struct KmInfo {
  int iters;      // Lloyd iterations run
  int done;       // 0 running, 1 strict convergence (labels repeated), 2 centre shift <= tol
  int empty;      // a cluster lost all its rows
  int changed;    // scratch: some label differs from the previous iteration
};

__global__ void km_center_kernel(const double* __restrict__ X, long ldx, const double* __restrict__ mean, int n, int d,
                                 double* __restrict__ Xc) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * d) return;
  const int r = (int)(gid / d), c = (int)(gid - (long)r * d);
  Xc[gid] = X[(long)r * ldx + c] - mean[c];
}

__global__ void km_csq_kernel(const double* __restrict__ C, int k, int d, double* __restrict__ csq) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double s = 0.0;
  for (int c = 0; c < d; ++c) s += C[(long)j * d + c] * C[(long)j * d + c];
  csq[j] = s;
}

// E step for one chunk of 256 rows + the chunk's partial sums of the M step.  LDS: centres [k][d], sums [k][d], labels.
__global__ __launch_bounds__(KM_CHUNK) void km_assign_kernel(const double* __restrict__ Xc, int n, int d, int k,
                                                            const double* __restrict__ C, const double* __restrict__ csq,
                                                            int* __restrict__ labels, const int* __restrict__ labels_old,
                                                            double* __restrict__ psum, int* __restrict__ pcnt,
                                                            KmInfo* __restrict__ info, int want_sums) {
  extern __shared__ __attribute__((aligned(16))) double km_lds[];
  if (info->done) return;  // converged in an earlier iteration of this batch of launches
  double* sC = km_lds;                  // [k][d]
  double* sS = km_lds + (long)k * d;    // [k][d]
  int* sL = reinterpret_cast<int*>(sS + (long)k * d);  // [KM_CHUNK]
  const int t = threadIdx.x, r0 = blockIdx.x * KM_CHUNK;
  for (int e = t; e < k * d; e += KM_CHUNK) {
    sC[e] = C[e];
    sS[e] = 0.0;
  }
  __syncthreads();
  const int row = r0 + t;
  int lab = -1;
  if (row < n) {
    const double* x = Xc + (long)row * d;
    double best = 0.0;
    for (int j = 0; j < k; ++j) {
      double dot = 0.0;
      for (int c = 0; c < d; ++c) dot = fma(x[c], sC[(long)j * d + c], dot);
      const double dist = csq[j] - 2.0 * dot;
      if (j == 0 || dist < best) {  // strict <: the first minimum wins, as in sklearn
        best = dist;
        lab = j;
      }
    }
    labels[row] = lab;
    if (labels_old && labels_old[row] != lab) info->changed = 1;  // benign race: everybody writes 1
  }
  sL[t] = lab;
  __syncthreads();
  if (!want_sums) return;
  // M step partials: thread c owns column c and walks the chunk's rows in order (deterministic)
  const int rows = min(KM_CHUNK, n - r0);
  for (int c = t; c < d; c += KM_CHUNK) {
    for (int r = 0; r < rows; ++r) {
      const int j = sL[r];
      sS[(long)j * d + c] += Xc[(long)(r0 + r) * d + c];
    }
  }
  for (int j = t; j < k; j += KM_CHUNK) {  // k may exceed the chunk (k <= 1024, k * d <= 8192)
    int cnt = 0;
    for (int r = 0; r < rows; ++r) cnt += (sL[r] == j);
    pcnt[(long)blockIdx.x * k + j] = cnt;
  }
  __syncthreads();
  for (int e = t; e < k * d; e += KM_CHUNK) psum[(long)blockIdx.x * k * d + e] = sS[e];
}

// The same E step + M-step partials with the chunk's rows staged through LDS (round 4).  The kernel above gives every thread a
// row and lets it read that row from global memory once per centre: 64 different cache lines per load instruction, the row
// fetched k times -- 223 us per call at n = 10,000, d = 128, k = 8, ~50 calls per window, 4 % of the benchmark's kernel time
// for 10 MFLOP.  Here a 256-row chunk travels as 256 / TR sub-tiles of TR rows, loaded coalesced into LDS (pitch d + 1);
// PT = 256 / TR threads share a row (centres j = part, part + PT, ...: every dot product still runs over c = 0 .. d - 1 in
// sequence, so the distances have the same bits) and agree on the first minimum by a lexicographic (distance, j) exchange;
// the M-step partials add the rows of the chunk in the same order as before, from LDS.  Results are bit-identical to the
// kernel above (same sums in the same order).
template <int TR>
__global__ __launch_bounds__(KM_CHUNK) void km_assign_tiled_kernel(const double* __restrict__ Xc, int n, int d, int k,
                                                                  const double* __restrict__ C, const double* __restrict__ csq,
                                                                  int* __restrict__ labels, const int* __restrict__ labels_old,
                                                                  double* __restrict__ psum, int* __restrict__ pcnt,
                                                                  KmInfo* __restrict__ info, int want_sums) {
  constexpr int PT = KM_CHUNK / TR;
  extern __shared__ __attribute__((aligned(16))) double km_lds[];
  if (info->done) return;  // converged in an earlier iteration of this batch of launches
  const int dp = d + 1;
  double* sC = km_lds;                       // [k][dp]
  double* sS = sC + (long)k * dp;            // [k][d]
  double* xs = sS + (long)k * d;             // [TR][dp]
  int* sL = reinterpret_cast<int*>(xs + (long)TR * dp);  // [KM_CHUNK]
  const int t = threadIdx.x, r0 = blockIdx.x * KM_CHUNK;
  for (int e = t; e < k * d; e += KM_CHUNK) {
    const int j = e / d, c = e - j * d;
    sC[(long)j * dp + c] = C[e];
    sS[e] = 0.0;
  }
  const int rows = min(KM_CHUNK, n - r0);
  const int row = t / PT, part = t % PT;
  for (int sub = 0; sub * TR < rows; ++sub) {
    const int rb = r0 + sub * TR, nr = min(TR, n - rb);
    __syncthreads();  // the centres are staged / the previous sub-tile is consumed
    for (int e = t; e < nr * d; e += KM_CHUNK) {
      const int r = e / d, c = e - r * d;
      xs[(long)r * dp + c] = Xc[(long)(rb + r) * d + c];
    }
    __syncthreads();
    double best = 1.7976931348623157e308;
    int lab = 0x7fffffff;
    if (row < nr) {
      const double* x = xs + (long)row * dp;
      for (int j = part; j < k; j += PT) {
        const double* cj = sC + (long)j * dp;
        double dot = 0.0;
        for (int c = 0; c < d; ++c) dot = fma(x[c], cj[c], dot);
        const double dist = csq[j] - 2.0 * dot;
        if (lab == 0x7fffffff || dist < best) {  // strict <: the first minimum wins, as in sklearn
          best = dist;
          lab = j;
        }
      }
    }
#pragma unroll
    for (int o = 1; o < PT; o <<= 1) {  // the PT threads of a row are adjacent lanes
      const double ob = __shfl_xor(best, o);
      const int ol = __shfl_xor(lab, o);
      if (ol != 0x7fffffff && (lab == 0x7fffffff || ob < best || (ob == best && ol < lab))) {
        best = ob;
        lab = ol;
      }
    }
    if (part == 0 && row < nr) {
      labels[rb + row] = lab;
      if (labels_old && labels_old[rb + row] != lab) info->changed = 1;  // benign race: everybody writes 1
      sL[sub * TR + row] = lab;
    }
    __syncthreads();
    if (want_sums) {  // M-step partials: thread c owns column c and walks the sub-tile's rows in order (deterministic)
      for (int c = t; c < d; c += KM_CHUNK)
        for (int r = 0; r < nr; ++r) sS[(long)sL[sub * TR + r] * d + c] += xs[(long)r * dp + c];
    }
  }
  if (!want_sums) return;
  __syncthreads();
  for (int j = t; j < k; j += KM_CHUNK) {  // k may exceed the chunk (k <= 1024, k * d <= 8192)
    int cnt = 0;
    for (int r = 0; r < rows; ++r) cnt += (sL[r] == j);
    pcnt[(long)blockIdx.x * k + j] = cnt;
  }
  for (int e = t; e < k * d; e += KM_CHUNK) psum[(long)blockIdx.x * k * d + e] = sS[e];
}

// M step: centres from the chunk partials (chunks in order), total squared shift, stopping rule.  One workgroup.
__global__ __launch_bounds__(1024) void km_update_kernel(const double* __restrict__ psum, const int* __restrict__ pcnt,
                                                        int nchunk, int k, int d, double* __restrict__ C,
                                                        double* __restrict__ csq, double tol, int first_iter,
                                                        KmInfo* __restrict__ info) {
  __shared__ double red[1024];
  __shared__ int s_cnt[1024];
  if (info->done) return;
  const int t = threadIdx.x;
  for (int j = t; j < k; j += 1024) {
    int cnt = 0;
    for (int ch = 0; ch < nchunk; ++ch) cnt += pcnt[(long)ch * k + j];
    s_cnt[j] = cnt;
    if (cnt == 0) info->empty = 1;
  }
  __syncthreads();
  double acc = 0.0;  // this thread's share of the squared shift, elements in a fixed order
  for (int e = t; e < k * d; e += 1024) {
    const int j = e / d;
    double s = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) s += psum[(long)ch * k * d + e];
    // sklearn averages by multiplying with the reciprocal (_k_means_common.pyx _average_centers: alpha = 1.0 / weight)
    const double nw = s_cnt[j] > 0 ? s * (1.0 / (double)s_cnt[j]) : C[e];
    const double df = nw - C[e];
    acc += df * df;
    C[e] = nw;
  }
  red[t] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  for (int j = t; j < k; j += 1024) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += C[(long)j * d + c] * C[(long)j * d + c];
    csq[j] = s;
  }
  if (t == 0) {
    info->iters += 1;
    if (!first_iter && info->changed == 0) info->done = 1;       // labels repeated: strict convergence
    else if (red[0] <= tol) info->done = 2;                       // centres stopped moving
    info->changed = 0;
  }
}

}  // namespace mused

using namespace mused;

extern "C" {

// workspace bytes for mused_kmeans_lloyd
long mused_kmeans_ws_bytes(int n, int d, int k) {
  if (n <= 0 || d <= 0 || k <= 0) return -1;
  const long nchunk = (n + KM_CHUNK - 1) / KM_CHUNK;
  return 8l * n * d + 8l * nchunk * k * d + 4l * nchunk * k + 8l * k + 8l * n + 4096;
}

// Replaces the Lloyd iterations of KMeans(n_clusters = k, random_state = seed).fit_predict(X)
// (matrix_operations.py:149-153) for the embedding X (n x d fp64, pitch ld, DEVICE), given -- all computed on the host
// exactly as scikit-learn does -- the column means `mean` (d), the k-means++ centres of the CENTRED rows `centers`
// (k x d, in: seeds, out: final centres of the centred data) and tol = mean(var(X, axis = 0)) * 1e-4.
// labels_out: n int32 (DEVICE).  info_out (4 ints, HOST): {iterations, 1 strict / 2 tol / 0 max_iter, empty-cluster
// flag, 0}.  BLOCKING (reads its stopping flag every few iterations).  k * d <= 8192.
int mused_kmeans_lloyd(const double* X, long ld, int n, int d, int k, const double* mean, double* centers, double tol,
                       int max_iter, int* labels_out, int* info_out, void* ws, long ws_bytes, void* stream) {
  MUSED_REQUIRE(X && mean && centers && labels_out && info_out && ws && n > 0 && d > 0 && k > 0 && k <= n && ld >= d,
                "mused_kmeans_lloyd: bad arguments");
  MUSED_REQUIRE((long)k * d <= 8192 && k <= 1024, "mused_kmeans_lloyd: k * d must be <= 8192");
  MUSED_REQUIRE(ws_bytes >= mused_kmeans_ws_bytes(n, d, k), "mused_kmeans_lloyd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = cdiv(n, KM_CHUNK);
  char* w = (char*)ws;
  double* Xc = (double*)w; w += 8l * n * d;
  double* psum = (double*)w; w += 8l * nchunk * k * d;
  double* csq = (double*)w; w += 8l * k;
  int* pcnt = (int*)w; w += 4l * nchunk * k;
  w = (char*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
  int* lab2 = (int*)w; w += 4l * n;
  w = (char*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
  KmInfo* info = (KmInfo*)w;
  const size_t lds_plain = 8 * 2 * (size_t)k * d + 4 * (KM_CHUNK + (size_t)k) + 16;
  // staged variant: the largest sub-tile (64 / 32 / 16 rows) whose LDS need fits; 0 = the row-per-thread kernel
  auto tiled_lds = [&](int tr) -> size_t { return 8 * ((size_t)k * (d + 1) + (size_t)k * d + (size_t)tr * (d + 1)) + 4 * KM_CHUNK + 16; };
  constexpr size_t LDS_MAX = 156 * 1024;
  static const bool plain_only = [] {
    const char* e = getenv("MUSED_KMEANS_ASSIGN");
    return e && e[0] == 'p';
  }();
  const int tr = plain_only ? 0 : (tiled_lds(64) <= LDS_MAX ? 64 : (tiled_lds(32) <= LDS_MAX ? 32 : (tiled_lds(16) <= LDS_MAX ? 16 : 0)));
  const size_t lds = tr ? tiled_lds(tr) : lds_plain;
  static std::once_flag once;
  static hipError_t aerr = hipSuccess;
  std::call_once(once, [] {
    aerr = hipFuncSetAttribute(reinterpret_cast<const void*>(km_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               8 * 2 * 8192 + 4 * (KM_CHUNK + 1024) + 16);
    if (aerr == hipSuccess)
      aerr = hipFuncSetAttribute(reinterpret_cast<const void*>(km_assign_tiled_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (aerr == hipSuccess)
      aerr = hipFuncSetAttribute(reinterpret_cast<const void*>(km_assign_tiled_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (aerr == hipSuccess)
      aerr = hipFuncSetAttribute(reinterpret_cast<const void*>(km_assign_tiled_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
  });
  MUSED_CHECK_HIP(aerr);
  auto assign = [&](int* cur_, const int* old_, int want) {
    switch (tr) {
      case 64: hipLaunchKernelGGL(km_assign_tiled_kernel<64>, dim3(nchunk), dim3(KM_CHUNK), lds, st, Xc, n, d, k, centers, csq, cur_, old_, psum, pcnt, info, want); break;
      case 32: hipLaunchKernelGGL(km_assign_tiled_kernel<32>, dim3(nchunk), dim3(KM_CHUNK), lds, st, Xc, n, d, k, centers, csq, cur_, old_, psum, pcnt, info, want); break;
      case 16: hipLaunchKernelGGL(km_assign_tiled_kernel<16>, dim3(nchunk), dim3(KM_CHUNK), lds, st, Xc, n, d, k, centers, csq, cur_, old_, psum, pcnt, info, want); break;
      default: hipLaunchKernelGGL(km_assign_kernel, dim3(nchunk), dim3(KM_CHUNK), lds, st, Xc, n, d, k, centers, csq, cur_, old_, psum, pcnt, info, want);
    }
  };
  MUSED_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(KmInfo), st));
  hipLaunchKernelGGL(km_center_kernel, dim3(cdiv((long)n * d, 256)), dim3(256), 0, st, X, ld, mean, n, d, Xc);
  hipLaunchKernelGGL(km_csq_kernel, dim3(cdiv(k, 64)), dim3(64), 0, st, centers, k, d, csq);
  KmInfo h;
  memset(&h, 0, sizeof(h));
  int* cur = labels_out;
  int* old = lab2;
  int it = 0;
  while (it < max_iter && !h.done) {
    const int batch = (max_iter - it) < 4 ? (max_iter - it) : 4;  // iterations between two reads of the stopping flag
    for (int b = 0; b < batch; ++b, ++it) {
      assign(cur, it > 0 ? old : (const int*)nullptr, 1);
      hipLaunchKernelGGL(km_update_kernel, dim3(1), dim3(1024), 0, st, psum, pcnt, nchunk, k, d, centers, csq, tol,
                         it == 0 ? 1 : 0, info);
      int* tmp = cur; cur = old; old = tmp;  // `old` now holds the labels of the iteration just queued
    }
    MUSED_LAUNCH_CHECK();
    MUSED_CHECK_HIP(hipMemcpyAsync(&h, info, sizeof(h), hipMemcpyDeviceToHost, st));
    MUSED_CHECK_HIP(hipStreamSynchronize(st));
    if (h.empty) break;
  }
  // labels of the last iteration that RAN are in one of the two buffers: iterations queued after convergence returned at
  // once, so parity of h.iters tells which.  Iteration i (0-based) wrote labels_out when i is even.
  int* last = ((h.iters - 1) % 2 == 0) ? labels_out : lab2;
  if (h.done != 1 && !h.empty) {
    // not strictly converged: labels must match the final centres (one more E step, no update)
    MUSED_CHECK_HIP(hipMemsetAsync(&info->done, 0, sizeof(int), st));
    assign(labels_out, (const int*)nullptr, 0);
    MUSED_LAUNCH_CHECK();
  } else if (last != labels_out) {
    MUSED_CHECK_HIP(hipMemcpyAsync(labels_out, last, 4l * n, hipMemcpyDeviceToDevice, st));
  }
  MUSED_CHECK_HIP(hipStreamSynchronize(st));
  info_out[0] = h.iters; info_out[1] = h.done; info_out[2] = h.empty; info_out[3] = 0;
  return MUSED_OK;
}

}  // extern "C"
