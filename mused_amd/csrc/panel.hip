// Tall-skinny panel factorizations used as the power-iteration normalisers of the
// randomized range finder (sklearn:utils/extmath.py:330-355):
//   * LU with partial pivoting, returning P*L in place (scipy.linalg.lu(..., permute_l=True))
//   * Householder QR, returning the economic Q (scipy.linalg.qr(..., mode="economic"))
// Panels are n x r row-major fp64 with n ~ 10^4, r <= ~300: a few MFLOP of work per column,
// strictly sequential over columns.  Each column step is a pair/triple of small kernels on
// the same stream (a kernel boundary is the cheapest grid-wide sync on this chip); the
// whole sequence is replayed from a hipGraph by the caller.
//
// Pivot rule: largest |a_ij| among not-yet-pivoted rows, ties to the smallest ORIGINAL row
// index (LAPACK's idamax breaks ties by current position, which only differs for exactly
// equal magnitudes).  Rows are never physically swapped: P*L is what the caller wants.
#include "internal.h"

namespace mused {

constexpr int PANEL_ROWS_PER_WG = 512;

// ---------------------------------------------------------------- LU -------------
constexpr int LU_ROWS_PER_WG = 16;

// partial pivot candidates of column jc: per workgroup (16 rows) the largest |a| among unpivoted rows
// (ties: smaller row).  A single workgroup scanning a strided column costs ~15 us at n = 10^4 (one
// cache line per element through one CU); spread over the update grid it is free.
__device__ __forceinline__ void lu_partial_colmax(const double* __restrict__ Y, int n, long ld, int jc,
                                                  const int* __restrict__ pivstep, double* __restrict__ pval,
                                                  int* __restrict__ pidx) {
  __shared__ double sv[LU_ROWS_PER_WG];
  __shared__ int si[LU_ROWS_PER_WG];
  const int lr = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int row = blockIdx.x * LU_ROWS_PER_WG + lr;
  if (tx == 0) {
    double a = -1.0;
    if (row < n && pivstep[row] == 0) a = fabs(Y[(long)row * ld + jc]);
    sv[lr] = a;
    si[lr] = row;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = sv[0];
    int ix = si[0];
    for (int w = 1; w < LU_ROWS_PER_WG; ++w)
      if (sv[w] > b) { b = sv[w]; ix = si[w]; }  // rows ascend with w: first maximum = smallest row
    pval[blockIdx.x] = b;
    pidx[blockIdx.x] = ix;
  }
}

__global__ __launch_bounds__(256) void lu_colmax_kernel(const double* __restrict__ Y, int n, long ld, int jc,
                                                       const int* __restrict__ pivstep, double* __restrict__ pval,
                                                       int* __restrict__ pidx) {
  lu_partial_colmax(Y, n, ld, jc, pivstep, pval, pidx);
}

// reduce the per-workgroup candidates, mark the pivot row, copy it to prow
__global__ __launch_bounds__(1024) void lu_pivot_kernel(const double* __restrict__ Y, int npart, int r, long ld, int j,
                                                       const double* __restrict__ pval, const int* __restrict__ pidx,
                                                       int* __restrict__ pivstep, double* __restrict__ prow) {
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_win;
  double best = -2.0;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < npart; i += 1024) {
    const double a = pval[i];
    const int ix = pidx[i];
    if (a > best || (a == best && ix < bi)) { best = a; bi = ix; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = s_val[0];
    int ix = s_idx[0];
    for (int w = 1; w < 16; ++w)
      if (s_val[w] > b || (s_val[w] == b && s_idx[w] < ix)) { b = s_val[w]; ix = s_idx[w]; }
    s_win = ix;
    pivstep[ix] = j + 1;
  }
  __syncthreads();
  const int win = s_win;
  for (int c = j + threadIdx.x; c < r; c += 1024) prow[c] = Y[(long)win * ld + c];
}

// rows not yet pivoted: l = a_ij * (1 / pivot); a_ij <- l; a_ic -= l * prow[c]  (c > j); then the
// workgroup's pivot candidate for column j + 1
__global__ __launch_bounds__(256) void lu_update_kernel(double* __restrict__ Y, int n, int r, long ld, int j, int k,
                                                       const int* __restrict__ pivstep,
                                                       const double* __restrict__ prow, double* __restrict__ pval,
                                                       int* __restrict__ pidx) {
  const int row = blockIdx.x * LU_ROWS_PER_WG + (threadIdx.x >> 4);
  const int tx = threadIdx.x & 15;
  if (row < n && pivstep[row] == 0) {
    double* y = Y + (long)row * ld;
    const double piv = prow[j];
    const double a = y[j];
    const double l = (piv != 0.0) ? a * (1.0 / piv) : a;
    for (int c = j + 1 + tx; c < r; c += 16) y[c] -= l * prow[c];
    if (tx == 0) y[j] = l;
  }
  if (j + 1 < k) {
    __syncthreads();  // column j + 1 of this workgroup's rows is final (written by lanes of the same waves)
    lu_partial_colmax(Y, n, ld, j + 1, pivstep, pval, pidx);
  }
}

// pivot rows: unit diagonal at their step, zeros to the right; keep only k = min(n, r) columns
__global__ void lu_finalize_kernel(double* __restrict__ Y, int n, int k, long ld, const int* __restrict__ pivstep) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int tx = threadIdx.x & 15;
  if (row >= n) return;
  const int s = pivstep[row] - 1;
  if (s < 0) return;
  double* y = Y + (long)row * ld;
  for (int c = s + tx; c < k; c += 16) y[c] = (c == s) ? 1.0 : 0.0;
}

// ---- blocked variant: LU_NB columns per launch pair -------------------------------------------------
// Same arithmetic, element by element and in the same order, as the column-at-a-time kernels above (the
// results are bit-identical), organised so that nothing walks a strided column of the row-major panel:
//   lu_trail_kernel  (n / 16 workgroups, rows in parallel): applies the LU_NB rank-1 updates of the finished
//                    panel to the rest of each unpivoted row, stores the row's multipliers, and EXPORTS the next
//                    LU_NB columns to a compact column-major buffer pc[c][row];
//   lu_panel_kernel  (ONE workgroup, 1024 threads): reads pc (coalesced, 40 doubles per thread in registers),
//                    eliminates its LU_NB columns with workgroup barriers only (pivot search, scaling, in-panel
//                    updates), writes the multipliers back to pc, and finishes the LU_NB pivot rows to the right
//                    of the panel (triangular solve) into prowN for the next lu_trail_kernel.
// 2 launches per 4 columns instead of 8, one pass over the trailing matrix instead of 4.
constexpr int LU_NB = 4;

template <int RPT>
__global__ __launch_bounds__(1024) void lu_panel_kernel(const double* __restrict__ Y, int n, int r, long ld, int j0,
                                                       int nbe, int* __restrict__ pivstep, double* __restrict__ pc,
                                                       double* __restrict__ prowN) {
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_win[LU_NB];
  __shared__ double s_prow[LU_NB][LU_NB];  // [c][c2 >= c]: panel entries of the pivot row of panel column c
  __shared__ double s_l[LU_NB][LU_NB];     // [s][t < s]: multipliers of pivot row s at the earlier panel columns
  __shared__ double s_inv;
  const int t = threadIdx.x;
  double a[RPT][LU_NB];
  bool act[RPT], act0[RPT];
#pragma unroll
  for (int ri = 0; ri < RPT; ++ri) {
    const int row = t + 1024 * ri;
    act[ri] = row < n && pivstep[row] == 0;
    act0[ri] = act[ri];
#pragma unroll
    for (int c = 0; c < LU_NB; ++c) a[ri][c] = (act[ri] && c < nbe) ? pc[(long)c * n + row] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < LU_NB; ++c) {
    if (c < nbe) {  // uniform
      double best = -2.0;
      int bi = 0x7fffffff;
#pragma unroll
      for (int ri = 0; ri < RPT; ++ri) {
        const double v = fabs(a[ri][c]);
        if (act[ri] && v > best) { best = v; bi = t + 1024 * ri; }  // rows ascend with ri: first maximum = smallest row
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if ((t & 63) == 0) { s_val[t >> 6] = best; s_idx[t >> 6] = bi; }
      __syncthreads();
      if (t == 0) {
        double b = s_val[0];
        int ix = s_idx[0];
        for (int w = 1; w < 16; ++w)
          if (s_val[w] > b || (s_val[w] == b && s_idx[w] < ix)) { b = s_val[w]; ix = s_idx[w]; }
        s_win[c] = ix;
        pivstep[ix] = j0 + c + 1;
      }
      __syncthreads();
      const int win = s_win[c];
#pragma unroll
      for (int ri = 0; ri < RPT; ++ri) {
        if (t + 1024 * ri == win) {
#pragma unroll
          for (int c2 = 0; c2 < LU_NB; ++c2) {
            if (c2 >= c) s_prow[c][c2] = a[ri][c2];
            else s_l[c][c2] = a[ri][c2];
          }
          s_inv = 1.0 / a[ri][c];  // one IEEE divide by the owner of the pivot row, not 1024
          act[ri] = false;
        }
      }
      __syncthreads();
      const double piv = s_prow[c][c];
      const double inv = s_inv;
#pragma unroll
      for (int ri = 0; ri < RPT; ++ri) {
        if (act[ri]) {
          const double l = (piv != 0.0) ? a[ri][c] * inv : a[ri][c];
          a[ri][c] = l;
#pragma unroll
          for (int c2 = c + 1; c2 < LU_NB; ++c2) a[ri][c2] -= l * s_prow[c][c2];
        }
      }
    }
  }
#pragma unroll
  for (int ri = 0; ri < RPT; ++ri) {
    if (act0[ri]) {
      const int row = t + 1024 * ri;
#pragma unroll
      for (int c = 0; c < LU_NB; ++c)
        if (c < nbe) pc[(long)c * n + row] = a[ri][c];
    }
  }
  // the finished pivot rows to the right of the panel (triangular solve with the unit-lower L11)
  for (int cc = j0 + nbe + t; cc < r; cc += 1024) {
    double u[LU_NB];
#pragma unroll
    for (int sgl = 0; sgl < LU_NB; ++sgl) {
      if (sgl < nbe) {
        double v = Y[(long)s_win[sgl] * ld + cc];
#pragma unroll
        for (int t2 = 0; t2 < sgl; ++t2) v -= s_l[sgl][t2] * u[t2];
        u[sgl] = v;
        prowN[(long)sgl * r + cc] = v;
      }
    }
  }
}

// Finished panel (j0p, nbp; nbp = 0: none yet) -> rows: store the panel entries, apply its rank-1 updates in
// column order to the right of it, export the next panel (j0p + nbp, nbn; nbn = 0: none left) to pc.
__global__ __launch_bounds__(256) void lu_trail_kernel(double* __restrict__ Y, int n, int r, long ld, int j0p, int nbp,
                                                      int nbn, const int* __restrict__ pivstep,
                                                      double* __restrict__ pc, const double* __restrict__ prowN) {
  const int row = blockIdx.x * LU_ROWS_PER_WG + (threadIdx.x >> 4);
  const int tx = threadIdx.x & 15;
  if (row >= n) return;
  const int ps = pivstep[row];
  double* y = Y + (long)row * ld;
  const bool in_panel = nbp > 0 && ps > j0p && ps <= j0p + nbp;  // one of the pivot rows of the finished panel
  if (ps != 0 && !in_panel) return;                             // pivoted earlier: final
  double l[LU_NB];
#pragma unroll
  for (int sgl = 0; sgl < LU_NB; ++sgl) l[sgl] = (sgl < nbp) ? pc[(long)sgl * n + row] : 0.0;
  if (tx < nbp) y[j0p + tx] = l[tx];
  if (in_panel) return;  // multipliers (and U entries lu_finalize overwrites) stored; the rest of the row is not used
  const int j0n = j0p + nbp;
  for (int c = j0n + tx; c < r; c += 16) {
    double v = y[c];
#pragma unroll
    for (int sgl = 0; sgl < LU_NB; ++sgl)
      if (sgl < nbp) v -= l[sgl] * prowN[(long)sgl * r + c];
    if (nbp > 0) y[c] = v;
    if (c - j0n < nbn) pc[(long)(c - j0n) * n + row] = v;  // first round of the loop only: lanes 0 .. nbn - 1
  }
}

// ---- one launch per column (MUSED_LU_MODE=2; measured in round 2, NOT the default) ---------------------------------
// Measured at n = 10^4, r = 138: 7.3 us per column launch (latency of its four dependent phases, not bandwidth) ->
// eigenstep 22.6 ms instead of 24.8 alone, but 313 workgroups x 1,380 launches per window compete with the sketch
// kernels for workgroup slots: 149 k rows/s instead of 152 k in the benchmark.  Kept for the stand-alone case.
// The blocked variant above funnels every panel through ONE workgroup: 640 KB per 4 columns through one CU (26 us
// alone, ~47 us beside the sketch kernels, whose workgroups leave no CU free for a 1024-thread workgroup).  Here a
// column costs ONE launch of n / 32 small workgroups and nothing is ever single-workgroup:
//   * every workgroup reduces the per-workgroup pivot candidates of column j itself (n / 32 (value, row) pairs = a few
//     KB from L2; same tie rule: largest |a|, then smallest row) -- the redundant reduce replaces the pivot launch;
//   * it loads the pivot row (r - j doubles) into LDS, eliminates its own 32 rows -- same expressions, element by element,
//     as lu_update_kernel: results are bit-identical to the other two variants --, and writes its candidate for column
//     j + 1 into the OTHER candidate buffer (workgroups of this launch may still be reading column j's).
constexpr int LUC_ROWS = 32;  // rows per workgroup, 8 threads per row

__device__ __forceinline__ void luc_candidates(const double* __restrict__ Y, int n, long ld, int jc,
                                               const int* __restrict__ pivstep, int skip_row, double* __restrict__ pval,
                                               int* __restrict__ pidx) {
  // workgroup's best unpivoted |a| of column jc (ties: smaller row); LDS scratch is declared by the caller's scope
  __shared__ double sv[LUC_ROWS];
  __shared__ int si[LUC_ROWS];
  const int lr = threadIdx.x >> 3, tx = threadIdx.x & 7;
  const int row = blockIdx.x * LUC_ROWS + lr;
  if (tx == 0) {
    double a = -1.0;
    if (row < n && row != skip_row && pivstep[row] == 0) a = fabs(Y[(long)row * ld + jc]);
    sv[lr] = a;
    si[lr] = row;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = sv[0];
    int ix = si[0];
    for (int w = 1; w < LUC_ROWS; ++w)
      if (sv[w] > b) { b = sv[w]; ix = si[w]; }  // rows ascend with w: first maximum = smallest row
    pval[blockIdx.x] = b;
    pidx[blockIdx.x] = ix;
  }
}

__global__ __launch_bounds__(256) void luc_first_kernel(const double* __restrict__ Y, int n, long ld,
                                                       const int* __restrict__ pivstep, double* __restrict__ pval,
                                                       int* __restrict__ pidx) {
  luc_candidates(Y, n, ld, 0, pivstep, -1, pval, pidx);
}

__global__ __launch_bounds__(256) void luc_step_kernel(double* __restrict__ Y, int n, int r, long ld, int j, int k,
                                                      int npart, int* __restrict__ pivstep, double* __restrict__ pval,
                                                      int* __restrict__ pidx) {
  extern __shared__ double s_prow[];  // [r - j]
  __shared__ double s_val[4];
  __shared__ int s_idx[4];
  __shared__ int s_win;
  const double* cv = pval + (long)(j & 1) * npart;
  const int* ci = pidx + (long)(j & 1) * npart;
  double best = -2.0;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < npart; i += 256) {
    const double a = cv[i];
    const int ix = ci[i];
    if (a > best || (a == best && ix < bi)) { best = a; bi = ix; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = s_val[0];
    int ix = s_idx[0];
    for (int w = 1; w < 4; ++w)
      if (s_val[w] > b || (s_val[w] == b && s_idx[w] < ix)) { b = s_val[w]; ix = s_idx[w]; }
    s_win = ix;
  }
  __syncthreads();
  const int win = s_win;
  for (int c = j + threadIdx.x; c < r; c += 256) s_prow[c - j] = Y[(long)win * ld + c];  // the pivot row is never updated
  const int row = blockIdx.x * LUC_ROWS + (threadIdx.x >> 3);
  const int tx = threadIdx.x & 7;
  const bool mine = row < n && row != win && pivstep[row] == 0;  // read before the owner marks `win` (a different row anyway)
  __syncthreads();
  if (win / LUC_ROWS == (int)blockIdx.x && threadIdx.x == 0) pivstep[win] = j + 1;
  if (mine) {
    double* y = Y + (long)row * ld;
    const double piv = s_prow[0];
    const double a = y[j];
    const double l = (piv != 0.0) ? a * (1.0 / piv) : a;
    for (int c = j + 1 + tx; c < r; c += 8) y[c] -= l * s_prow[c - j];
    if (tx == 0) y[j] = l;
  }
  if (j + 1 < k) {
    __syncthreads();  // column j + 1 of this workgroup's rows is final (written by lanes of this workgroup)
    luc_candidates(Y, n, ld, j + 1, pivstep, win, pval + (long)((j + 1) & 1) * npart, pidx + (long)((j + 1) & 1) * npart);
  }
}

// ws_f64_len: doubles available behind prow (the blocked path needs LU_NB * (r + n))
int lu_permute_l(double* Y, int n, int r, long ld, int* pivstep, double* prow, long ws_f64_len, hipStream_t st) {
  // workspace layout, column-at-a-time path: pivstep[n] ints, then npart ints of candidate rows; prow[r] doubles,
  // then npart doubles.  Blocked path: pivstep[n]; prowN[LU_NB][r], then pc[LU_NB][n].
  const int k = n < r ? n : r;
  const int npart = cdiv(n, LU_ROWS_PER_WG);
  int zrc = zero_ints(pivstep, n, st);
  if (zrc) return zrc;
  static const int lu_mode = [] {  // 0 column-at-a-time (2 launches per column), 1 blocked panels, 2 one launch per column
    const char* ub = getenv("MUSED_LU_BLOCKED");
    const char* um = getenv("MUSED_LU_MODE");
    if (um) return atoi(um);
    return (ub && ub[0] == '0') ? 0 : 1;
  }();
  const int npc = cdiv(n, LUC_ROWS);
  // candidate buffers of the one-launch-per-column path: 2 x npc doubles and 2 x npc ints behind prow[r]
  const bool percol = lu_mode == 2 && ws_f64_len >= (long)r + 3l * npc + 2;
  const bool blocked = !percol && lu_mode != 0 && n <= 10240 && ws_f64_len >= (long)LU_NB * ((long)r + n);
  if (percol) {
    double* pval = prow + r;
    int* pidx = reinterpret_cast<int*>(pval + 2l * npc);
    hipLaunchKernelGGL(luc_first_kernel, dim3(npc), dim3(256), 0, st, Y, n, ld, pivstep, pval, pidx);
    for (int j = 0; j < k; ++j)
      hipLaunchKernelGGL(luc_step_kernel, dim3(npc), dim3(256), sizeof(double) * (r - j), st, Y, n, r, ld, j, k, npc, pivstep,
                         pval, pidx);
  } else if (blocked) {
    double* prowN = prow;
    double* pc = prow + (long)LU_NB * r;
    int nbp = 0;
    for (int j0 = 0; j0 < k; j0 += LU_NB) {
      const int nbe = (k - j0) < LU_NB ? (k - j0) : LU_NB;
      hipLaunchKernelGGL(lu_trail_kernel, dim3(npart), dim3(256), 0, st, Y, n, r, ld, j0 - nbp, nbp, nbe, pivstep, pc, prowN);
      if (n <= 3072)
        hipLaunchKernelGGL(lu_panel_kernel<3>, dim3(1), dim3(1024), 0, st, Y, n, r, ld, j0, nbe, pivstep, pc, prowN);
      else
        hipLaunchKernelGGL(lu_panel_kernel<10>, dim3(1), dim3(1024), 0, st, Y, n, r, ld, j0, nbe, pivstep, pc, prowN);
      nbp = nbe;
    }
    const int j0l = ((k - 1) / LU_NB) * LU_NB;
    hipLaunchKernelGGL(lu_trail_kernel, dim3(npart), dim3(256), 0, st, Y, n, r, ld, j0l, nbp, 0, pivstep, pc, prowN);
  } else {
    int* pidx = pivstep + n;
    double* pval = prow + r;
    hipLaunchKernelGGL(lu_colmax_kernel, dim3(npart), dim3(256), 0, st, Y, n, ld, 0, pivstep, pval, pidx);
    for (int j = 0; j < k; ++j) {
      hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(1024), 0, st, Y, npart, r, ld, j, pval, pidx, pivstep, prow);
      hipLaunchKernelGGL(lu_update_kernel, dim3(npart), dim3(256), 0, st, Y, n, r, ld, j, k, pivstep, prow, pval, pidx);
    }
  }
  hipLaunchKernelGGL(lu_finalize_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, Y, n, k, ld, pivstep);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// ---------------------------------------------------------------- QR -------------
// Column j: Householder vector of Y[j:, j] (v_j = 1 implied, stored explicitly), LAPACK dlarfg convention:
// beta = -sign(alpha) * hypot(alpha, |x|), tau = (beta - alpha)/beta, v = x / (alpha - beta); tau[j] = 0 when
// x == 0.  The one-workgroup kernel never walks the strided column of the row-major panel: the update of
// column j - 1 (rows in parallel) exports column j to the contiguous buffer `col`, this kernel turns it into
// the contiguous Householder vector `vcol`, and the update of column j writes v back into Y.
// run_if (optional, device): every kernel of the Householder chain returns at once unless *run_if != 0 -- the chain is
// recorded into the eigenstep graph as the FALLBACK of the Cholesky-QR factorisation (rsvd.hip)
__global__ __launch_bounds__(1024) void qr_house_kernel(const double* __restrict__ col, int n, int j,
                                                       double* __restrict__ tau, double* __restrict__ vcol,
                                                       const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  __shared__ double s_sum[16];
  __shared__ double s_scale;
  double s = 0.0;
  for (int i = j + 1 + threadIdx.x; i < n; i += 1024) {
    const double v = col[i];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double xn2 = 0.0;
    for (int w = 0; w < 16; ++w) xn2 += s_sum[w];
    const double alpha = col[j];
    if (xn2 == 0.0) {
      tau[j] = 0.0;
      s_scale = 0.0;
    } else {
      const double beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
      tau[j] = (beta - alpha) / beta;
      s_scale = 1.0 / (alpha - beta);
    }
    vcol[j] = 1.0;
  }
  __syncthreads();
  const double sc = s_scale;
  for (int i = j + 1 + threadIdx.x; i < n; i += 1024) vcol[i] = col[i] * sc;
}

// col[i] = Y[i][j]  (rows in parallel; only for the first column, later ones come out of qr_update_kernel)
__global__ void qr_export_col_kernel(const double* __restrict__ Y, int n, long ld, int j, double* __restrict__ col,
                                     const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) col[i] = Y[(long)i * ld + j];
}

// wpart[chunk][c] = sum_{i in chunk, i >= j} v_i * T[i][c]   for c in [c_lo, c_hi)
// (v = vcol if given, else column j of Y; T is Y itself during factorisation, the Q accumulator afterwards)
__global__ __launch_bounds__(1024) void qr_dot_kernel(const double* __restrict__ Y, long ldy, int j,
                                                     const double* __restrict__ vcol, const double* __restrict__ T,
                                                     long ldt, int n, int c_lo, int c_hi, double* __restrict__ wpart,
                                                     int wld, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  __shared__ double red[16][64];
  const int chunk = blockIdx.x;
  const int r0 = max(j, chunk * PANEL_ROWS_PER_WG);
  const int r1 = min(n, (chunk + 1) * PANEL_ROWS_PER_WG);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cb = c_lo; cb < c_hi; cb += 64) {
    const int c = cb + tx;
    double s = 0.0;
    if (c < c_hi)
      for (int i = r0 + ty; i < r1; i += 16) s += (vcol ? vcol[i] : Y[(long)i * ldy + j]) * T[(long)i * ldt + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < c_hi) {
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) acc += red[w][tx];
      wpart[(long)chunk * wld + c] = acc;
    }
    __syncthreads();
  }
}

// T[i][c] -= tau_j * v_i * w[c],  w[c] = sum_chunk wpart[chunk][c]   (rows i >= j).
// Factorisation phase (vcol given, T == Y): also Y[i][j] <- v_i and col[i] <- the updated T[i][j + 1].
__global__ __launch_bounds__(256) void qr_update_kernel(double* __restrict__ Y, long ldy, int j,
                                                       const double* __restrict__ vcol, double* __restrict__ col,
                                                       double* __restrict__ T, long ldt, int n, int c_lo, int c_hi,
                                                       const double* __restrict__ wpart, int wld, int nchunk,
                                                       int first_chunk, const double* __restrict__ tau,
                                                       const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  extern __shared__ double w[];  // [c_hi - c_lo]
  const double tj = tau[j];
  if (tj == 0.0 && !vcol) return;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += 256) {
    double s = 0.0;
    for (int k = first_chunk; k < nchunk; ++k) s += wpart[(long)k * wld + c];
    w[c - c_lo] = s * tj;
  }
  __syncthreads();
  const int row = j + blockIdx.x * 16 + (threadIdx.x >> 4);
  const int tx = threadIdx.x & 15;
  if (row >= n) return;
  const double v = vcol ? vcol[row] : Y[(long)row * ldy + j];
  double* t = T + (long)row * ldt;
  if (vcol && tx == 15) Y[(long)row * ldy + j] = v;
  for (int c = c_lo + tx; c < c_hi; c += 16) {
    double x = t[c];
    if (tj != 0.0) {
      x -= v * w[c - c_lo];
      t[c] = x;
    }
    if (col && c == j + 1) col[row] = x;
  }
}

__global__ void set_identity_kernel(double* __restrict__ Q, int n, int r, long ld, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * r) return;
  const int row = (int)(gid / r), c = (int)(gid - (long)row * r);
  Q[(long)row * ld + c] = (row == c) ? 1.0 : 0.0;
}

int qr_economic(double* Y, int n, int r, long ldy, double* Q, long ldq, double* tau, double* wpart,
                hipStream_t st, const int* run_if) {
  // requires n >= r.  wpart: ceil(n/512) * r partial sums, then col[n], then vcol[n]
  const int nchunk = cdiv(n, PANEL_ROWS_PER_WG);
  const int wld = r;
  double* col = wpart + (long)nchunk * r;
  double* vcol = col + n;
  hipLaunchKernelGGL(qr_export_col_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, Y, n, ldy, 0, col, run_if);
  for (int j = 0; j < r; ++j) {
    hipLaunchKernelGGL(qr_house_kernel, dim3(1), dim3(1024), 0, st, col, n, j, tau, vcol, run_if);
    const int first = j / PANEL_ROWS_PER_WG;
    if (j + 1 < r)
      hipLaunchKernelGGL(qr_dot_kernel, dim3(nchunk), dim3(1024), 0, st, Y, ldy, j, vcol, Y, ldy, n, j + 1, r, wpart, wld,
                         run_if);
    // (the last column has nothing to update: c_lo = c_hi = r; the launch only writes v back into Y)
    hipLaunchKernelGGL(qr_update_kernel, dim3(cdiv(n - j, 16)), dim3(256), sizeof(double) * (r - j), st, Y, ldy, j, vcol,
                       col, Y, ldy, n, j + 1, r, wpart, wld, nchunk, first, tau, run_if);
  }
  hipLaunchKernelGGL(set_identity_kernel, dim3(cdiv((long)n * r, 256)), dim3(256), 0, st, Q, n, r, ldq, run_if);
  for (int j = r - 1; j >= 0; --j) {
    const int first = j / PANEL_ROWS_PER_WG;
    hipLaunchKernelGGL(qr_dot_kernel, dim3(nchunk), dim3(1024), 0, st, Y, ldy, j, (const double*)nullptr, Q, ldq, n, j,
                       r, wpart, wld, run_if);
    hipLaunchKernelGGL(qr_update_kernel, dim3(cdiv(n - j, 16)), dim3(256), sizeof(double) * (r - j), st, Y, ldy, j,
                       (const double*)nullptr, (double*)nullptr, Q, ldq, n, j, r, wpart, wld, nchunk, first, tau, run_if);
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // namespace mused

using namespace mused;

extern "C" {

// In place: Y (n x r, ld) <- P*L of its LU factorisation with partial pivoting, first
// min(n, r) columns (scipy.linalg.lu(Y, permute_l=True)[0]).
// ws_int: n + ceil(n/16) ints, ws_f64: 4 * (r + n) doubles.
int mused_lu_permute_l(double* Y, int n, int r, long ld, int* ws_int, double* ws_f64, void* stream) {
  MUSED_REQUIRE(Y && ws_int && ws_f64 && n > 0 && r > 0 && ld >= r, "mused_lu_permute_l: bad arguments");
  return lu_permute_l(Y, n, r, ld, ws_int, ws_f64, 4l * ((long)r + n), (hipStream_t)stream);
}

// Q (n x r, ldq) <- economic Householder QR of Y (n x r, ldy; destroyed).  n >= r.
// ws_f64: r + ceil(n/512)*r + 2 n doubles.
int mused_qr_economic(double* Y, int n, int r, long ldy, double* Q, long ldq, double* ws_f64, void* stream) {
  MUSED_REQUIRE(Y && Q && ws_f64 && n >= r && r > 0 && ldy >= r && ldq >= r, "mused_qr_economic: need n >= r");
  return qr_economic(Y, n, r, ldy, Q, ldq, ws_f64, ws_f64 + r, (hipStream_t)stream);
}

}  // extern "C"
