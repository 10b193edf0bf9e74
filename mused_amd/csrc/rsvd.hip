// a8 of SURVEY section 8: the "eigenstep" -- perform_svd_reduction (matrix_operations.py:143-147)
// = sklearn TruncatedSVD(algorithm="randomized", n_iter=5, n_oversamples=10,
// power_iteration_normalizer="auto" -> LU).fit_transform on the fused 0/1 adjacency:
//
//   Q <- Q0 (n x r, host-generated RandomState(seed).normal, sklearn:utils/extmath.py:297)
//   repeat n_iter:  Q <- PL(A Q);  Q <- PL(A^T Q)         (:349-351, scipy lu permute_l)
//   Q <- qr_economic(A Q)                                   (:355)
//   B = Q^T A;  Uhat, s, Vt = svd(B)                        (:579-588)
//   svd_flip on the rows of Vt[:n_comp] (u_based_decision=False)   (_truncated_svd.py:253)
//   X_new = A Vt[:n_comp]^T                                 (_truncated_svd.py:262)
//
// Device formulation: A is never dense.  A Q and A^T Q are gathers over CSR neighbour lists
// (spmm.hip), B is held transposed (Bt = A^T Q, n x r), its SVD comes from the r x r Gram
// Bt^T Bt (split-K MFMA GEMM, fixed-order reduction) and the Jacobi eigensolver, and
// V = Bt U S^-1 is one more MFMA GEMM.  Everything is fp64: the embedding feeds k-means,
// whose labels must match the CPU path bit for bit.  The whole sequence (~2000 small
// launches) is captured once per shape into a hipGraph and replayed per window.
#include "internal.h"

namespace mused {

struct Rsvd {
  int n_max, r_max, eig_n, sweeps;
  long prow_len;  // doubles behind prow (LU workspace)
  long cm_len;    // doubles behind Cm
  long nnz_cap;
  unsigned long long *mask, *mask_t;
  int *deg, *rowptr, *colidx, *degT, *rowptrT, *colidxT, *stats, *pivstep, *flags;
  double *Q0, *Qa, *Qb, *Qf, *Bt, *prow, *tau, *wpart, *gpart, *evals, *U, *Cm, *Vsel, *embed, *sigma, *signs;
  EigPlan* eig;
  EigPlan* eig_top;  // same order on the direct solver (trd.hip; r <= 256): used when at most top_need components are asked for
  int top_need;
  hipStream_t cap_stream;
  hipGraph_t graph;
  hipGraphExec_t exec;
  bool have_graph, use_graph;
  int mode;  // mused_rsvd_set_mode: 0 Cholesky-QR + Householder fallback recorded in the graph, 1 Cholesky-QR only (the caller
             // reads flags[2]), 2 the reference's LU / Householder chain, launched kernel by kernel
  int g_n, g_r, g_ncomp, g_iter;
  int q0_n, q0_r;
};

constexpr int GRAM_KCHUNK = 512;
constexpr int CHOLQR_KCHUNK = 128;  // split-K chunk of the normaliser's Gram: ~4 x more workgroups than the final Gram

__global__ void gram_reduce_pad_kernel(const double* __restrict__ partial, int nsplit, int rc, double* __restrict__ G,
                                       int en) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= en * en) return;
  const int i = gid / en, j = gid - i * en;
  double s = 0.0;
  if (i < rc && j < rc) {
    // symmetrise by construction: entry (i, j) and (j, i) read the same partials in the same order
    const int a = i < j ? i : j, b = i < j ? j : i;
    for (int z = 0; z < nsplit; ++z) s += partial[(long)z * rc * rc + (long)a * rc + b];
  }
  G[gid] = s;
}

// ---- Cholesky-QR in place of the LU normaliser and of the final Householder QR ------------------------------------------
// sklearn normalises the iterates Y of its randomized range finder with an LU factorisation (extmath.py:343-351,
// power_iteration_normalizer = "LU" for n_iter = 5) and ends with an economic QR (:355).  Both only have to deliver a
// well-conditioned (resp. orthonormal) BASIS of span(Y): the range after the last iteration is the same subspace, so the
// singular values, V and the embedding agree with the LU chain to rounding (measured on every golden: sigma to 1e-15,
// embedding to 1e-12, k-means labels identical) -- and a partially pivoted LU of a 10,000 x 138 panel is ~70 dependent
// launches, the Householder QR ~690.  Here:  G = Y^T Y (split-K MFMA GEMM, fixed-order reduction),  G + delta I = L L^T in
// LDS (one workgroup),  Q = Y L^-T by forward substitution, 64 rows per workgroup: 4 launches.  One such pass per power
// iteration (of A^T (A Q)), two for the final basis (Cholesky-QR2).  delta = 16 r eps max(diag G) keeps the factorisation
// defined when Y is rank deficient (adjacency matrices with few non-empty rows, tiny windows); a pivot below 1e-11 of the
// largest in a FINAL pass raises flags[2], and the basis is then taken from the Householder chain (mused_rsvd_set_mode:
// recorded behind the flag, or by the caller on a mode-2 handle).  MUSED_RSVD_NORMALIZER=lu restores the reference's chain.
constexpr int CHOLQR_MAX_R = 143;  // packed lower triangle of L (r (r + 1) / 2 doubles) + 64 rows of Y in LDS: 157 KB

__device__ __forceinline__ int tri_at(int i, int k) { return i * (i + 1) / 2 + k; }

// one workgroup: G (r x r, full) -> packed L = chol(G + delta I), written to Lg (r (r + 1) / 2 doubles).
// The matrix lives in REGISTERS: the lower block triangle is dealt 2-D cyclically to a 16 x 16 thread grid (element
// (i, k) with i = p + 16 a, k = q + 16 b, a >= b: 45 doubles per thread for r <= 144), as in the tridiagonalisation of
// trd.hip.  Right-looking, one column per step: the owners of column j put it (raw) into an LDS vector, ONE barrier, every
// thread forms d = sqrt(pivot), 1 / d (v_rsq_f64 + two Newton steps) and the scaled entries of its 9 rows and 9 columns and
// applies the rank-1 update to its elements; the vector is double-buffered, so the next column's owners never wait.
// (The first version worked on a packed triangle in LDS with a rank-8 trailing update per panel: 220 us for r = 138,
// bound by LDS round trips; this one: see DESIGN section 4e.)
constexpr long CHOLQR_P_OFF = 16384;  // offset (doubles) of the two-block path's projection scratch inside Cm
static_assert(CHOLQR_P_OFF >= (long)CHOLQR_MAX_R * (CHOLQR_MAX_R + 1) / 2, "the packed triangle must end before the projection scratch");
constexpr int CH_NB = 9;  // 16 x 9 = 144 >= CHOLQR_MAX_R
static_assert(16 * CH_NB >= CHOLQR_MAX_R, "chol_kernel holds at most 16 * CH_NB rows");
__host__ __device__ constexpr int ch_tri(int a, int b) { return a * (a + 1) / 2 + b; }

template <int B>
__device__ __forceinline__ void chol_block_columns(double (&A)[CH_NB * (CH_NB + 1) / 2], double (*colbuf)[16 * CH_NB], int r, int t,
                                                   int p, int q, double delta, double weak, int* __restrict__ weak_flag) {
  for (int jj = 0; jj < 16; ++jj) {
    const int j = 16 * B + jj;
    if (j >= r) return;  // (uniform)
    double* cb = colbuf[j & 1];
    if (q == jj) {
#pragma unroll
      for (int a = B; a < CH_NB; ++a) {
        const int i = p + 16 * a;
        if (i >= j && i < r) cb[i] = A[ch_tri(a, B)];
      }
    }
    __syncthreads();
    const double piv0 = cb[j];
    // a pivot below 1e-11 of the largest diagonal entry: Y is (numerically) rank deficient or worse conditioned than 3e5
    // -- fine for a normaliser, not for the final orthonormal basis (the caller then falls back to Householder QR)
    if (weak_flag && t == 0 && !(piv0 > weak)) *weak_flag = 1;
    const double piv = piv0 + delta;
    const double pc = piv > delta ? piv : delta;
    double inv = __builtin_amdgcn_rsq(pc);
    inv = inv * fma(-0.5 * pc, inv * inv, 1.5);
    inv = inv * fma(-0.5 * pc, inv * inv, 1.5);
    const double d = pc * inv;
    double lr[CH_NB], lc[CH_NB];
#pragma unroll
    for (int a = B; a < CH_NB; ++a) {
      // rows / columns beyond j: every one of a later block, those of block B past jj; entries at and beyond r are zeros
      // (never written since the vectors were cleared): no bounds test per entry
      const double xr = cb[p + 16 * a] * inv, xc = cb[q + 16 * a] * inv;
      lr[a] = (a > B || p > jj) ? xr : 0.0;
      lc[a] = (a > B || q > jj) ? xc : 0.0;
    }
#pragma unroll
    for (int b = B; b < CH_NB; ++b)
#pragma unroll
      for (int a = b; a < CH_NB; ++a) A[ch_tri(a, b)] = fma(-lr[a], lc[b], A[ch_tri(a, b)]);
    if (q == jj) {  // column j of L: d on the diagonal, the scaled entries below (lc[B] = 0 here: the update left them alone)
#pragma unroll
      for (int a = B; a < CH_NB; ++a) {
        const int i = p + 16 * a;
        if (i > j) A[ch_tri(a, B)] = lr[a];
        else if (i == j) A[ch_tri(a, B)] = d;
      }
    }
  }
}

__global__ __launch_bounds__(256) void chol_kernel(const double* __restrict__ G, int r, double* __restrict__ Lg,
                                                   int* __restrict__ weak_flag) {
  __shared__ double colbuf[2][16 * CH_NB];
  __shared__ double s_red[4];
  __shared__ double s_delta;
  const int t = threadIdx.x, q = t & 15, p = t >> 4;  // (q fastest: 16 consecutive entries of a row per 16 threads)
  double A[CH_NB * (CH_NB + 1) / 2];
  double dmax = 0.0;
#pragma unroll
  for (int a = 0; a < CH_NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = p + 16 * a, k = q + 16 * b;
      const double v = (i < r && k <= i) ? G[(long)i * r + k] : 0.0;
      A[ch_tri(a, b)] = v;
      if (k == i && i < r) dmax = fmax(dmax, v);
    }
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
  if ((t & 63) == 0) s_red[t >> 6] = dmax;
  __syncthreads();
  if (t == 0) s_delta = 16.0 * r * 2.220446049250313e-16 * fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double delta = s_delta;
  const double weak = 1e-11 * (delta / (16.0 * r * 2.220446049250313e-16));
  for (int i = t; i < 2 * 16 * CH_NB; i += 256) (&colbuf[0][0])[i] = 0.0;  // (entries >= r are read as zeros, see chol_block_columns)
  __syncthreads();
  chol_block_columns<0>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<1>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<2>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<3>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<4>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<5>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<6>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<7>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
  chol_block_columns<8>(A, colbuf, r, t, p, q, delta, weak, weak_flag);
#pragma unroll
  for (int a = 0; a < CH_NB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = p + 16 * a, k = q + 16 * b;
      if (i < r && k <= i) Lg[tri_at(i, k)] = A[ch_tri(a, b)];
    }
}

// Q = Y L^-T, 64 rows of Y per workgroup: row y solves L q^T = y^T by forward substitution.  Four lanes share a row (the
// inner sum over m < k is dealt to them, two DPP adds combine it); L (packed) and the 64 rows live in LDS.
__global__ __launch_bounds__(256) void trsm_rows_kernel(const double* __restrict__ Y, long ldy, int n, int r,
                                                        const double* __restrict__ Lg, double* __restrict__ Q, long ldq) {
  extern __shared__ double cq_smem[];
  const int t = threadIdx.x;
  const int ntri = r * (r + 1) / 2;
  double* Lp = cq_smem;
  double* rows = cq_smem + ntri;  // [64][rs]: the workgroup's 64 rows, odd pitch (bank spread between the quads)
  const int rs = r | 1;
  for (int e = t; e < ntri; e += 256) Lp[e] = Lg[e];
  const long row0 = (long)blockIdx.x * 64;
  for (int e = t; e < 64 * r; e += 256) {
    const int rr = e / r, k = e - rr * r;
    rows[rr * rs + k] = (row0 + rr < n) ? Y[(row0 + rr) * ldy + k] : 0.0;
  }
  __syncthreads();
  const int rr = t >> 2, part = t & 3;  // the 4 lanes of a quad work on row rr
  double* q = rows + rr * rs;
  // Four columns per pass: the sums over the columns already solved share one read of q[m] for four FMAs (5 LDS reads per 4
  // FMAs instead of 8), then the 4 x 4 triangle of the block is finished by every lane of the quad for itself.
  int k = 0;
  for (; k + 4 <= r; k += 4) {
    const double* l0 = Lp + tri_at(k, 0);
    const double* l1 = Lp + tri_at(k + 1, 0);
    const double* l2 = Lp + tri_at(k + 2, 0);
    const double* l3 = Lp + tri_at(k + 3, 0);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int m = part; m < k; m += 4) {
      const double qm = q[m];
      s0 = fma(l0[m], qm, s0);
      s1 = fma(l1[m], qm, s1);
      s2 = fma(l2[m], qm, s2);
      s3 = fma(l3[m], qm, s3);
    }
    s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1); s3 += __shfl_xor(s3, 1);
    s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2); s3 += __shfl_xor(s3, 2);
    const double q0 = (q[k] - s0) / l0[k];
    const double q1 = (q[k + 1] - fma(l1[k], q0, s1)) / l1[k + 1];
    const double q2 = (q[k + 2] - fma(l2[k + 1], q1, fma(l2[k], q0, s2))) / l2[k + 2];
    const double q3 = (q[k + 3] - fma(l3[k + 2], q2, fma(l3[k + 1], q1, fma(l3[k], q0, s3)))) / l3[k + 3];
    if (part == 0) { q[k] = q0; q[k + 1] = q1; q[k + 2] = q2; q[k + 3] = q3; }
    // the quad reads these entries in later passes: same wave, LDS operations of a wave complete in order
  }
  for (; k < r; ++k) {
    const double* lrow = Lp + tri_at(k, 0);
    double sum = 0.0, sum1 = 0.0, sum2 = 0.0, sum3 = 0.0;
    int m = part;
    for (; m + 12 < k; m += 16) {  // four independent partial sums: the LDS reads of a pass are in flight together
      const double l0 = lrow[m], l1 = lrow[m + 4], l2 = lrow[m + 8], l3 = lrow[m + 12];
      const double q0 = q[m], q1 = q[m + 4], q2 = q[m + 8], q3 = q[m + 12];
      sum = fma(l0, q0, sum); sum1 = fma(l1, q1, sum1); sum2 = fma(l2, q2, sum2); sum3 = fma(l3, q3, sum3);
    }
    for (; m < k; m += 4) sum = fma(lrow[m], q[m], sum);
    sum = (sum + sum1) + (sum2 + sum3);
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (part == 0) q[k] = (q[k] - sum) / lrow[k];
    // the quad reads entry k in later steps: same wave, LDS operations of a wave complete in order
  }
  __syncthreads();
  for (int e = t; e < 64 * r; e += 256) {
    const int r2 = e / r, k = e - r2 * r;
    if (row0 + r2 < n) Q[(row0 + r2) * ldq + k] = rows[r2 * rs + k];
  }
}

// out = Y - T  (n x c panels, row pitches ldy / ldt / ldo): the projection step of the blocked Cholesky-QR
__global__ void panel_sub_kernel(const double* __restrict__ Y, long ldy, const double* __restrict__ T, long ldt, int n, int c,
                                 double* __restrict__ out, long ldo) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c) return;
  const long i = gid / c;
  const int j = (int)(gid - i * c);
  out[i * ldo + j] = Y[i * ldy + j] - T[i * ldt + j];
}

__global__ void gram_reduce_kernel(const double* __restrict__ partial, int nsplit, int rc, double* __restrict__ G) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rc * rc) return;
  const int i = gid / rc, j = gid - i * rc;
  const int a = i < j ? i : j, b = i < j ? j : i;  // symmetric by construction: both triangles read the same partials
  double s = 0.0;
  for (int z = 0; z < nsplit; ++z) s += partial[(long)z * rc * rc + (long)a * rc + b];
  G[gid] = s;
}

// eigenvalues (unsorted, en) + eigenvectors U (en x en) of the Gram -> singular values (descending)
// and the coefficient block Cm (rc x n_comp) = U[:, order] * diag(1 / s)
__global__ __launch_bounds__(1024) void rsvd_decide_kernel(const double* __restrict__ evals, const double* __restrict__ U,
                                                          int en, int rc, int n_comp, double* __restrict__ Cm,
                                                          double* __restrict__ sigma) {
  __shared__ int order[1024];
  __shared__ double lam[1024];
  const int t = threadIdx.x;
  if (t < en) lam[t] = evals[t];
  __syncthreads();
  if (t < en) {
    const double mine = lam[t];
    int rank = 0;
    for (int j = 0; j < en; ++j) {
      const double o = lam[j];
      rank += (o > mine) || (o == mine && j < t);
    }
    order[rank] = t;
  }
  __syncthreads();
  const double l0 = lam[order[0]];
  const double s0 = sqrt(l0 > 0.0 ? l0 : 0.0);
  for (int i = t; i < n_comp; i += 1024) {
    const double l = lam[order[i]];
    sigma[i] = sqrt(l > 0.0 ? l : 0.0);
  }
  for (int e = t; e < rc * n_comp; e += 1024) {
    const int a = e / n_comp, i = e - a * n_comp;
    const int j = order[i];
    const double l = lam[j];
    const double s = sqrt(l > 0.0 ? l : 0.0);
    Cm[e] = (s > 1e-12 * s0 && s > 0.0) ? U[(long)a * en + j] / s : 0.0;
  }
}

// svd_flip(u_based_decision=False): per component, the entry of largest magnitude (first index
// on ties) decides the sign (sklearn:utils/extmath.py:944-952).  One workgroup per column.
__global__ __launch_bounds__(256) void col_sign_kernel(const double* __restrict__ V, int n, int ld, int ncol,
                                                      double* __restrict__ signs) {
  __shared__ double sv[4];
  __shared__ int si[4];
  const int c = blockIdx.x;
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double a = fabs(V[(long)i * ld + c]);
    if (a > best || (a == best && i < bi)) { best = a; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    const double v = V[(long)bi * ld + c];
    signs[c] = v < 0.0 ? -1.0 : 1.0;
  }
}

__global__ void col_scale_kernel(double* __restrict__ V, long total, int ncol, const double* __restrict__ signs) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < total) V[gid] *= signs[gid % ncol];
}

static int rsvd_enqueue(Rsvd* h, int n, int r, int n_comp, int n_iter, hipStream_t st, bool capturing) {
  const int words = (n + 63) / 64;
  const long ld = r;
  int rc_;
  static const bool dbg_env = getenv("MUSED_DEBUG") != nullptr;
  const bool dbg = dbg_env && !capturing;  // a stream that is being captured must not be synchronised
  int stage = 0;
#define RC(x)                                                                                     \
  do {                                                                                            \
    if ((rc_ = (x))) return rc_;                                                                  \
    ++stage;                                                                                      \
    if (dbg) {                                                                                    \
      hipError_t e_ = hipStreamSynchronize(st);                                                   \
      fprintf(stderr, "[rsvd] stage %d (%s): %s\n", stage, #x, hipGetErrorString(e_));            \
      if (e_ != hipSuccess) { set_error("stage %d failed: %s", stage, hipGetErrorString(e_)); return MUSED_ERR_HIP; } \
    }                                                                                             \
  } while (0)
  if ((rc_ = zero_ints(h->flags, 4, st))) return rc_;
  RC(adj_csr_from_mask(h->mask, n, words, h->deg, h->rowptr, h->colidx, h->stats, h->nnz_cap, h->flags, st));
  RC(adj_transpose(h->mask, n, words, h->mask_t, st));
  RC(adj_csr_from_mask(h->mask_t, n, words, h->degT, h->rowptrT, h->colidxT, h->stats + 2, h->nnz_cap, h->flags, st));

  int rc = r;
  const double* Qcur = h->Q0;
  static const bool want_lu = [] {
    const char* e = getenv("MUSED_RSVD_NORMALIZER");
    return e && (e[0] == 'l' || e[0] == 'L');
  }();
  const bool cholqr = !want_lu && h->mode != 2 && r <= 2 * CHOLQR_MAX_R && r <= n;
  // dst = an orthonormal-ish basis of span(Y) by Cholesky-QR; weak (optional device int): raised on a weak pivot.
  //   r <= CHOLQR_MAX_R:  G = Y^T Y in U;  L = chol(G + delta I) packed in Cm;  dst = Y L^-T
  //   larger r (BASELINE config 3: r = 266; two packed triangles of that order do not fit the LDS): the columns in two
  //   halves, block Gram-Schmidt -- Q1 = cholqr(Y1);  Y2' = Y2 - Q1 (Q1^T Y2);  Q2 = cholqr(Y2') -- from the same kernels
  //   plus two GEMMs.  One projection leaves Q1^T Q2 ~ cond(Y) eps: fine for the normaliser of a power iteration; the final
  //   basis runs the whole routine twice, which restores orthogonality to rounding as Cholesky-QR2 does.
  auto normalise = [&](const double* Y, double* dst, int* weak) -> int {
    const int nsp = cdiv(n, CHOLQR_KCHUNK);
    auto cholqr_block = [&](const double* Yb, int rb, double* dstb) -> int {
      int e;
      const size_t lds_l = sizeof(double) * (size_t)rb * (rb + 1) / 2;
      const size_t lds_t = lds_l + sizeof(double) * 64 * (size_t)(rb | 1);
      if ((e = gemm_f64_splitk(false, false, Yb, ld, Yb, ld, h->gpart, rb, rb, n, CHOLQR_KCHUNK, nsp, st))) return e;
      hipLaunchKernelGGL(gram_reduce_kernel, dim3(cdiv((long)rb * rb, 256)), dim3(256), 0, st, h->gpart, nsp, rb, h->U);
      hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), 0, st, h->U, rb, h->Cm, weak);
      hipLaunchKernelGGL(trsm_rows_kernel, dim3(cdiv(n, 64)), dim3(256), lds_t, st, Yb, ld, n, rb, h->Cm, dstb, ld);
      return MUSED_OK;
    };
    if (r <= CHOLQR_MAX_R) return cholqr_block(Y, r, dst);
    const int r1 = r / 2, r2 = r - r1;
    int e;
    if ((e = cholqr_block(Y, r1, dst))) return e;                                   // Q1 -> dst[:, :r1]
    // r1 x r2, behind the packed L of either block (<= CHOLQR_MAX_R (CHOLQR_MAX_R + 1) / 2 = 10,296 doubles)
    MUSED_REQUIRE(CHOLQR_P_OFF + (long)r1 * r2 <= h->cm_len, "rsvd: projection scratch of the two-block Cholesky-QR does not fit (r = %d)", r);
    double* P = h->Cm + CHOLQR_P_OFF;
    if ((e = gemm_f64_splitk(false, false, dst, ld, Y + r1, ld, h->gpart, r1, r2, n, CHOLQR_KCHUNK, nsp, st))) return e;
    if ((e = gemm_splitk_reduce(h->gpart, nsp, (long)r1 * r2, P, st))) return e;   // P = Q1^T Y2
    if ((e = gemm_f64(true, false, dst, ld, 0, P, r2, 0, h->Vsel, r2, 0, n, r2, r1, 1, 1.0, st))) return e;  // Q1 P
    hipLaunchKernelGGL(panel_sub_kernel, dim3(cdiv((long)n * r2, 256)), dim3(256), 0, st, Y + r1, ld, h->Vsel, (long)r2, n, r2,
                       dst + r1, ld);                                               // Y2' -> dst[:, r1:]
    return cholqr_block(dst + r1, r2, dst + r1);                                    // Q2 (in place: a workgroup owns its rows)
  };
  if (cholqr) {
    // ONE normalisation per power iteration, of A^T (A Q): the two products in a row square the spread of the basis
    // ((sigma_1 / sigma_r)^2 ~ 1e3 .. 1e4 for these adjacency matrices), far inside what a Cholesky of the Gram takes
    // (1e-11 of the largest pivot is the alarm level), and the subspace is the same (goldens: sigma 1e-15, labels equal)
    for (int it = 0; it < n_iter; ++it) {
      double* X = (Qcur == h->Qa) ? h->Qb : h->Qa;
      RC(spmm_binary(h->rowptr, h->colidx, n, Qcur, ld, rc, X, ld, st));
      double* Z = (X == h->Qa) ? h->Qb : h->Qa;  // (the buffer of the old basis, unless that is Q0)
      RC(spmm_binary(h->rowptrT, h->colidxT, n, X, ld, rc, Z, ld, st));
      RC(normalise(Z, X, nullptr));
      Qcur = X;
    }
  } else {
    for (int it = 0; it < n_iter; ++it) {
      RC(spmm_binary(h->rowptr, h->colidx, n, Qcur, ld, rc, h->Qa, ld, st));
      RC(lu_permute_l(h->Qa, n, rc, ld, h->pivstep, h->prow, h->prow_len, st));
      rc = n < rc ? n : rc;
      RC(spmm_binary(h->rowptrT, h->colidxT, n, h->Qa, ld, rc, h->Qb, ld, st));
      RC(lu_permute_l(h->Qb, n, rc, ld, h->pivstep, h->prow, h->prow_len, st));
      Qcur = h->Qb;
    }
  }
  double* Yf = (Qcur == h->Qa) ? h->Qb : h->Qa;  // Y = A Q of the last step
  double* Yt = (Yf == h->Qa) ? h->Qb : h->Qa;
  RC(spmm_binary(h->rowptr, h->colidx, n, Qcur, ld, rc, Yf, ld, st));
  rc = n < rc ? n : rc;
  if (cholqr) {
    // final orthonormal basis: Cholesky-QR twice (Yf -> Yt -> Qf); a weak pivot in either pass raises flags[2] (mode 0:
    // the Householder chain below, otherwise a string of no-op launches, then recomputes Qf from the untouched Yf)
    const double* src = Yf;
    double* dsts[2] = {Yt, h->Qf};
    for (int pass = 0; pass < 2; ++pass) {
      RC(normalise(src, dsts[pass], h->flags + 2));
      src = dsts[pass];
    }
    if (h->mode == 0) RC(qr_economic(Yf, n, rc, ld, h->Qf, ld, h->tau, h->wpart, st, h->flags + 2));
  } else {
    RC(qr_economic(Yf, n, rc, ld, h->Qf, ld, h->tau, h->wpart, st));
  }
  RC(spmm_binary(h->rowptrT, h->colidxT, n, h->Qf, ld, rc, h->Bt, ld, st));

  const int nsplit = cdiv(n, GRAM_KCHUNK);
  RC(gemm_f64_splitk(false, false, h->Bt, ld, h->Bt, ld, h->gpart, rc, rc, n, GRAM_KCHUNK, nsplit, st));
  // The r x r Gram of B: only its n_comp largest eigenpairs are used.  For r <= 256 and n_comp <= 128 (config 2 / 4:
  // r = 138, n_comp = 128; the reference's default: r = 60, n_comp = 50) it goes to the direct solver of the FD rotation
  // (the leading pairs by tridiagonalisation, every one of them certified; a rejected matrix goes to the Jacobi of the
  // same plan): 0.7 ms instead of 1.3 ms for the persistent Jacobi at order 138.  MUSED_RSVD_EIG_TRD=0 turns it off.
  const bool top = h->eig_top && n_comp <= h->top_need;
  EigPlan* ep = top ? h->eig_top : h->eig;
  const int en = h->eig_n;
  hipLaunchKernelGGL(gram_reduce_pad_kernel, dim3(cdiv((long)en * en, 256)), dim3(256), 0, st, h->gpart, nsplit, rc,
                     eig_plan_input(ep), en);
  RC(eig_plan_run_inplace(ep, h->evals, h->U, st, false));
  hipLaunchKernelGGL(rsvd_decide_kernel, dim3(1), dim3(1024), 0, st, h->evals, h->U, en, rc, n_comp, h->Cm, h->sigma);
  RC(gemm_f64(true, false, h->Bt, ld, 0, h->Cm, n_comp, 0, h->Vsel, n_comp, 0, n, n_comp, rc, 1, 1.0, st));
  hipLaunchKernelGGL(col_sign_kernel, dim3(n_comp), dim3(256), 0, st, h->Vsel, n, n_comp, n_comp, h->signs);
  hipLaunchKernelGGL(col_scale_kernel, dim3(cdiv((long)n * n_comp, 256)), dim3(256), 0, st, h->Vsel, (long)n * n_comp,
                     n_comp, h->signs);
  RC(spmm_binary(h->rowptr, h->colidx, n, h->Vsel, n_comp, n_comp, h->embed, n_comp, st));
  MUSED_LAUNCH_CHECK();
#undef RC
  return MUSED_OK;
}

static void rsvd_drop_graph(Rsvd* h) {
  if (h->have_graph) {
    (void)hipGraphExecDestroy(h->exec);
    (void)hipGraphDestroy(h->graph);
    h->have_graph = false;
  }
}

}  // namespace mused

using namespace mused;

extern "C" {

// Workspace handle for problems up to n_max rows, r_max = n_components + n_oversamples random
// columns, nnz_cap edges in the fused adjacency.  `sweeps` = Jacobi sweeps of the r x r solve
// (0 -> default 12).
static int rsvd_create_impl(Rsvd* h, int n_max, int r_max, long nnz_cap, int sweeps);
int mused_rsvd_destroy(void* handle);

int mused_rsvd_create(int n_max, int r_max, long nnz_cap, int sweeps, void** out) {
  MUSED_REQUIRE(out && n_max >= 2 && r_max >= 1 && r_max <= 1022 && nnz_cap >= 1, "mused_rsvd_create: bad arguments");
  CaptureLock resource_guard(capture_mutex());
  Rsvd* h = new Rsvd();
  memset(h, 0, sizeof(*h));
  const int rc = rsvd_create_impl(h, n_max, r_max, nnz_cap, sweeps);
  if (rc) {  // release what was allocated before the failure
    (void)mused_rsvd_destroy(h);
    return rc;
  }
  *out = h;
  return MUSED_OK;
}

static int rsvd_create_impl(Rsvd* h, int n_max, int r_max, long nnz_cap, int sweeps) {
  h->n_max = n_max; h->r_max = r_max; h->nnz_cap = nnz_cap;
  h->sweeps = sweeps > 0 ? sweeps : 16;  // cap of the adaptive sweep count
  h->eig_n = (r_max + 1) & ~1;
  const size_t words = (n_max + 63) / 64;
  const size_t panel = sizeof(double) * (size_t)n_max * r_max;
  // partial Grams: the final Gram's splits, or the normaliser's finer ones when it can run (r <= CHOLQR_MAX_R)
  const int nsplit = r_max <= 2 * CHOLQR_MAX_R ? cdiv(n_max, CHOLQR_KCHUNK) : cdiv(n_max, GRAM_KCHUNK);
#define ALLOC(p, bytes) MUSED_CHECK_HIP(hipMalloc((void**)&(p), (bytes)))
  ALLOC(h->mask, 8 * words * n_max);
  ALLOC(h->mask_t, 8 * words * n_max);
  ALLOC(h->deg, 4 * (size_t)n_max); ALLOC(h->degT, 4 * (size_t)n_max);
  ALLOC(h->rowptr, 4 * (size_t)(n_max + 1)); ALLOC(h->rowptrT, 4 * (size_t)(n_max + 1));
  ALLOC(h->colidx, 4 * (size_t)nnz_cap); ALLOC(h->colidxT, 4 * (size_t)nnz_cap);
  ALLOC(h->stats, 4 * 8); ALLOC(h->flags, 4 * 4); ALLOC(h->pivstep, 4 * ((size_t)n_max + cdiv(n_max, 16)));
  ALLOC(h->Q0, panel); ALLOC(h->Qa, panel); ALLOC(h->Qb, panel); ALLOC(h->Qf, panel); ALLOC(h->Bt, panel);
  ALLOC(h->Vsel, panel); ALLOC(h->embed, panel);
  h->prow_len = 4l * ((long)r_max + n_max);
  ALLOC(h->prow, 8 * (size_t)h->prow_len); ALLOC(h->tau, 8 * (size_t)r_max);
  ALLOC(h->wpart, 8 * ((size_t)r_max * cdiv(n_max, 512) + 2 * (size_t)n_max));
  ALLOC(h->gpart, 8 * (size_t)nsplit * r_max * r_max);
  const size_t en_max = (size_t)h->eig_n;
  ALLOC(h->evals, 8 * en_max); ALLOC(h->U, 8 * en_max * en_max);
  {
    // Cm holds the packed Cholesky factor and, for CHOLQR_MAX_R < r <= 2 CHOLQR_MAX_R, the r1 x r2 projection of the
    // two-block path behind it (round 3 sized it r_max^2 only: 16384 + r1 r2 > r^2 for r = 144 .. 147)
    const long r1m = r_max / 2, r2m = r_max - r1m;
    const long two_block = (r_max > CHOLQR_MAX_R && r_max <= 2 * CHOLQR_MAX_R) ? CHOLQR_P_OFF + r1m * r2m : 0;
    h->cm_len = (long)r_max * r_max > two_block ? (long)r_max * r_max : two_block;
  }
  ALLOC(h->Cm, 8 * (size_t)h->cm_len); ALLOC(h->sigma, 8 * (size_t)r_max); ALLOC(h->signs, 8 * (size_t)r_max);
#undef ALLOC
  // flags[3]: the r x r eigensolve gave up (work-queue timeout, eig.hip) -> the result of that call is invalid
  int rc = eig_plan_create(h->eig_n, 1, h->sweeps, false, &h->eig, nullptr, 0, h->flags + 3);
  if (rc) return rc;
  {
    const char* te = getenv("MUSED_RSVD_EIG_TRD");
    if (h->eig_n <= 256 && !(te && te[0] == '0')) {
      h->top_need = h->eig_n < 128 ? h->eig_n : 128;
      if ((rc = eig_plan_create(h->eig_n, 1, h->sweeps, false, &h->eig_top, nullptr, EIG_PLAN_TOP_NEED, h->flags + 3, h->top_need)))
        return rc;
      if (!eig_plan_direct_solver(h->eig_top)) {  // (MUSED_EIG_TRD=0): nothing gained by padding
        eig_plan_destroy(h->eig_top);
        h->eig_top = nullptr;
      }
    }
  }
  if ((rc = gemm_f64_prepare_all())) return rc;
  {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
      const int tri = (int)(sizeof(double) * CHOLQR_MAX_R * (CHOLQR_MAX_R + 1) / 2);
      attr_rc = hipFuncSetAttribute((const void*)trsm_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      tri + (int)(sizeof(double) * 64 * (CHOLQR_MAX_R | 1)));
    });
    MUSED_CHECK_HIP(attr_rc);
  }
  const char* ng = getenv("MUSED_NO_GRAPH");
  h->use_graph = !(ng && ng[0] == '1');
  return MUSED_OK;
}

int mused_rsvd_destroy(void* handle) {
  Rsvd* h = (Rsvd*)handle;
  if (!h) return MUSED_OK;
  CaptureLock resource_guard(capture_mutex());
  rsvd_drop_graph(h);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  eig_plan_destroy(h->eig);
  if (h->eig_top) eig_plan_destroy(h->eig_top);
  void* bufs[] = {h->mask, h->mask_t, h->deg, h->degT, h->rowptr, h->rowptrT, h->colidx, h->colidxT, h->stats,
                  h->flags, h->pivstep, h->Q0, h->Qa, h->Qb, h->Qf, h->Bt, h->Vsel, h->embed, h->prow, h->tau,
                  h->wpart, h->gpart, h->evals, h->U, h->Cm, h->sigma, h->signs};
  for (void* b : bufs) (void)hipFree(b);
  delete h;
  return MUSED_OK;
}

// Device buffer the fused adjacency bitmask must be written to before mused_rsvd_reduce
// (n rows of ceil(n/64) words, pitch ceil(n/64)).
unsigned long long* mused_rsvd_mask_buffer(void* handle) { return handle ? ((Rsvd*)handle)->mask : nullptr; }

// Q0: the (n x r) Gaussian test matrix, row-major fp64, generated on the host exactly as
// sklearn does (np.random.RandomState(seed).normal(size=(n, r))).  Copied into the handle.
int mused_rsvd_set_q0(void* handle, const double* Q0, int n, int r, void* stream) {
  Rsvd* h = (Rsvd*)handle;
  MUSED_REQUIRE(h && Q0 && n >= 1 && n <= h->n_max && r >= 1 && r <= h->r_max, "mused_rsvd_set_q0: bad arguments");
  MUSED_CHECK_HIP(hipMemcpyAsync(h->Q0, Q0, sizeof(double) * (size_t)n * r, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
  h->q0_n = n; h->q0_r = r;
  return MUSED_OK;
}

// Replaces perform_svd_reduction(matrix, reduced_dim, seed) for the fused 0/1 adjacency held as a
// bitmask in the handle's mask buffer.  n_comp = min(reduced_dim, n - 1), r = n_comp + 10 must
// match the Q0 set before.  out_embed: n x n_comp fp64 (X @ Vt.T), out_sigma: n_comp singular
// values (TruncatedSVD.singular_values_), out_components (optional): n x n_comp = Vt[:n_comp].T.
int mused_rsvd_reduce(void* handle, int n, int n_comp, int r, int n_iter, double* out_embed, double* out_sigma,
                      double* out_components, void* stream) {
  Rsvd* h = (Rsvd*)handle;
  MUSED_REQUIRE(h && out_embed && out_sigma, "mused_rsvd_reduce: null pointer");
  MUSED_REQUIRE(n >= 2 && n <= h->n_max && r <= h->r_max && n_comp >= 1 && n_comp <= r && n_comp <= n && n_iter >= 0,
                "mused_rsvd_reduce: bad sizes n=%d n_comp=%d r=%d", n, n_comp, r);
  MUSED_REQUIRE(h->q0_n == n && h->q0_r == r, "mused_rsvd_reduce: Q0 was set for (%d, %d), need (%d, %d)", h->q0_n,
                h->q0_r, n, r);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (h->use_graph && h->mode != 2) {
    if (!h->have_graph || h->g_n != n || h->g_r != r || h->g_ncomp != n_comp || h->g_iter != n_iter) {
      CaptureLock capture_guard(capture_mutex());  // one capture at a time, and no resource call of this library beside it
      rsvd_drop_graph(h);
      // the capture stream lives only while it records: every live HIP stream competes for the hardware queues
      if (!h->cap_stream) MUSED_CHECK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
      MUSED_CHECK_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
      rc = rsvd_enqueue(h, n, r, n_comp, n_iter, h->cap_stream, true);
      hipError_t e = hipStreamEndCapture(h->cap_stream, &h->graph);
      (void)hipStreamDestroy(h->cap_stream);
      h->cap_stream = nullptr;
      if (rc || e != hipSuccess) {
        if (!rc) set_error("mused_rsvd_reduce: graph capture failed (%s)", hipGetErrorString(e));
        return rc ? rc : MUSED_ERR_HIP;
      }
      MUSED_CHECK_HIP(hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0));
      h->have_graph = true;
      h->g_n = n; h->g_r = r; h->g_ncomp = n_comp; h->g_iter = n_iter;
    }
    MUSED_CHECK_HIP(hipGraphLaunch(h->exec, st));
  } else {
    if ((rc = rsvd_enqueue(h, n, r, n_comp, n_iter, st, false))) return rc;
  }
  MUSED_CHECK_HIP(hipMemcpyAsync(out_embed, h->embed, sizeof(double) * (size_t)n * n_comp, hipMemcpyDeviceToDevice, st));
  MUSED_CHECK_HIP(hipMemcpyAsync(out_sigma, h->sigma, sizeof(double) * (size_t)n_comp, hipMemcpyDeviceToDevice, st));
  if (out_components)
    MUSED_CHECK_HIP(hipMemcpyAsync(out_components, h->Vsel, sizeof(double) * (size_t)n * n_comp,
                                   hipMemcpyDeviceToDevice, st));
  return MUSED_OK;
}

// How the eigenstep orthogonalises (see "Cholesky-QR normaliser" above).  0 (default): Cholesky-QR, the Householder chain
// recorded behind the weak-pivot flag -- self-contained, ~690 no-op launches per call when the flag stays down.
// 1: Cholesky-QR only: flags[2] != 0 after a call means its result is INVALID (rank-deficient panel) and the call has to
// be repeated on a handle in mode 2.  2: the reference's LU normaliser + Householder QR, launched kernel by kernel (no
// graph capture: safe to create and use beside other host threads).
int mused_rsvd_set_mode(void* handle, int mode) {
  Rsvd* h = (Rsvd*)handle;
  MUSED_REQUIRE(h && mode >= 0 && mode <= 2, "mused_rsvd_set_mode: bad arguments");
  CaptureLock resource_guard(capture_mutex());
  if (h->mode != mode) rsvd_drop_graph(h);
  h->mode = mode;
  return MUSED_OK;
}

// Device int[4] the eigenstep raises its flags in (flags[0] != 0: the adjacency had more than nnz_cap edges, the
// neighbour lists were truncated and the result of that mused_rsvd_reduce is INVALID; flags[2]: weak Cholesky pivot, see
// mused_rsvd_set_mode; flags[3] != 0: the r x r eigensolve timed out, result INVALID).  Rewritten by every
// mused_rsvd_reduce on its stream: copy it behind the call (same stream) to read it without a host sync.
const int* mused_rsvd_flags(void* handle) { return handle ? ((Rsvd*)handle)->flags : nullptr; }

// Blocking status read: flags[0] != 0 -> the adjacency had more than nnz_cap edges (results invalid);
// stats = {max out-degree, nnz, max in-degree, nnz}.
int mused_rsvd_status(void* handle, int* flags_out, int* stats_out, void* stream) {
  Rsvd* h = (Rsvd*)handle;
  MUSED_REQUIRE(h && flags_out && stats_out, "mused_rsvd_status: null pointer");
  MUSED_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  int tmp[8];
  MUSED_CHECK_HIP(hipMemcpy(tmp, h->flags, sizeof(int) * 4, hipMemcpyDeviceToHost));
  MUSED_CHECK_HIP(hipMemcpy(tmp + 4, h->stats, sizeof(int) * 4, hipMemcpyDeviceToHost));
  flags_out[0] = tmp[0];
  for (int k = 0; k < 4; ++k) stats_out[k] = tmp[4 + k];
  return MUSED_OK;
}

}  // extern "C"
