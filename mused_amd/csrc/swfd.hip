// a5-a7 of SURVEY section 8: SeqBasedSWFD -- the sequence-based sliding-window Frequent
// Directions sketch mused imports from its (un-vendored) `swfd` submodule (main.py:10,62,67,70).
// The algorithm is this repo's specification (oracle/swfd_oracle.py):
// L = ceil(log2 R) + 1 dump-snapshot levels, theta_j = 2^j N / l, a MAIN and an AUX sketch per
// level; all 2L sketches rotate together every l rows of the current epoch (and at the epoch
// end), AUX replaces MAIN at every epoch start.
//
// Device formulation of one rotation ("the per-window SVD/rotation step"), batched over the
// S = 2L sketches, every sketch a (2l x d) fp64 buffer [kept rows ; pending rows ; zeros]:
//     G_s   = buf_s buf_s^T                    (2l x 2l)   MFMA fp64 GEMM, K = d
//     G_s   = U diag(lam) U^T                              batched Jacobi (eig.hip)
//     decide: sort lam, delta = lam[l-1], s2 = max(lam - delta, 0); rows with s2 >= theta_s are
//             dumped to the snapshot ring, the rest kept; Wc_s = diag(sqrt(s2/lam)) U[:, top l]^T
//     T_s   = Wc_s buf_s                       (l x d)     MFMA fp64 GEMM, K = 2l
//     scatter T_s rows to buf_s (kept) / snapshot ring (dumped), zero the tail.
// Control flow is data independent (the rotation schedule depends only on the row counter),
// all data-dependent decisions (dumps, expiry, level choice) stay on the device: no host sync
// anywhere on the update or query path.
#include <vector>

#include "internal.h"

namespace mused {

struct Swfd {
  int N, d, ell, L, S, cap, n2, n4, n3;
  int lanes;  // independent sketch sets advanced in lockstep (S = lanes * 2L sketches in every launch)
  double R;
  long i;    // rows seen so far
  int pend;  // rows appended since the last rotation (they sit raw in every buffer)
  long restart_mark;  // value of i for which the epoch-start swap has already been done
  int skip_dead;  // 1: MAIN sketches that can no longer be selected in their epoch are frozen (swfd_rep_kernel)
  bool twin;  // MAIN and AUX of every level still identical (both empty at the start of this epoch, no state imported)
  int sweeps;
  // persistent state
  double* buf;         // S x n2 x d
  double* queue;       // S x cap x d
  long long* qt;       // S x cap   snapshot timestamps
  int* meta;           // S x 4     {nk, qhead, qcount, dumps the next level did not make, since the last clear}
  int* rep;            // S         representative of the sketch's state class for the next rotation (see swfd_rep_kernel)
  long long* dropped;  // S         largest timestamp of a snapshot lost to the ring capacity
  double* theta;       // S
  // rotation workspace
  double *T, *Wc, *evals, *U;
  int gram_split;   // > 1: the buffers' Gram matrices by a batched split-K product (long rows: the K loop of a tile is the latency
  double* gpart;    //      of the rotation and the tiles of a launch do not fill the GPU) -- S x gram_split x n2 x n2 partial sums
  int* plan;      // S x ell x 2  {kind (0 none, 1 keep, 2 dump), position}
  int* keep_src;  // S x ell      buffer position -> row of T
  long long* now_dev;
  int* status;  // device word, sticky: bit 0 = an eigensolve of this sketch gave up (work-queue timeout): results invalid
  int* status_host;  // pinned copy target of mused_swfd_status (asynchronous copy on the caller's stream)
  EigPlan* eig;
  // query workspace
  double *stack, *evals_q, *Uq, *Wq, *Bout, *sig_out, *qinfo;
  int* qsel;  // {level, nsnap, nk, slots[cap]}
  EigPlan* eigq;
  EigPlan* eigq3;  // order n3 = 3l (even): enough whenever nothing is pending (snapshots <= 2l, kept rows <= l - 1)
  // pre-rotation of complete l-row input blocks (see swfd_prerotate)
  int pre_chunk;        // blocks per batch (0: off)
  double *pre_in, *pre_out, *pre_gram;  // [chunk][lanes][l][d] fp64 rows in / out, [chunk][lanes][l][l] Gram matrices
  EigPlan* eigp;        // order 2l, batch chunk * lanes, fixed sweeps, no sort
};

// sketch index = lane * 2L + kind * L + level  (kind 0 = MAIN, 1 = AUX)
static inline int sk_index(const Swfd* h, int lane, int level, int kind) { return lane * 2 * h->L + kind * h->L + level; }

// ---- update ------------------------------------------------------------------------
// Row sources: plain element rows (float / double / int64, pitch in elements) or BIT rows -- the rows of a 0/1
// adjacency held as a bitmask (pitch in 64-bit words), which is how the reference's SWFDMC approach feeds the
// sketch (main.py:65-67: one row of the fused W x W matrix per fit()) without the dense matrix ever existing.
struct BitWord { unsigned long long w; };
template <typename T>
__device__ __forceinline__ double row_elem(const T* row, int c) { return (double)row[c]; }
template <>
__device__ __forceinline__ double row_elem<BitWord>(const BitWord* row, int c) {
  return (double)((row[c >> 6].w >> (c & 63)) & 1ull);
}

template <typename T>
__global__ void swfd_append_kernel(const T* __restrict__ X, long ldx, long lane_stride, int per_lane, int m, int d,
                                   int n2, int pend, const int* __restrict__ meta, double* __restrict__ buf) {
  const int r = blockIdx.x, s = blockIdx.y;
  X += (long)(s / per_lane) * lane_stride;
  const int row = meta[s * 4 + 0] + pend + r;
  if (row >= n2) return;  // cannot happen by construction (nk <= l - 1, pend + m <= l)
  double* dst = buf + ((long)s * n2 + row) * d;
  const T* src = X + (long)r * ldx;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = row_elem<T>(src, c);
}

// epoch start: MAIN <- AUX, AUX <- empty  (sketch index = kind * L + level)
__global__ void swfd_restart_kernel(double* __restrict__ buf, double* __restrict__ queue, long long* __restrict__ qt,
                                    int* __restrict__ meta, long long* __restrict__ dropped, int L, long buf_elems,
                                    long queue_elems, int cap) {
  const int lane = blockIdx.y / L, j = blockIdx.y - lane * L;
  const int sm = lane * 2 * L + j, sa = sm + L;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < buf_elems) {
    buf[sm * buf_elems + gid] = buf[sa * buf_elems + gid];
    buf[sa * buf_elems + gid] = 0.0;
  }
  if (gid < queue_elems) queue[sm * queue_elems + gid] = queue[sa * queue_elems + gid];
  if (gid < cap) qt[sm * cap + gid] = qt[sa * cap + gid];
  if (gid == 0) {
    for (int k = 0; k < 4; ++k) {
      meta[sm * 4 + k] = meta[sa * 4 + k];
      meta[sa * 4 + k] = 0;
    }
    dropped[sm] = dropped[sa];
    dropped[sa] = 0;
  }
}

__global__ void swfd_set_now_kernel(long long* now, long long v) { *now = v; }

// Sketches of one kind (MAIN or AUX) of one lane see the same rows and differ only in the dump threshold
// theta_j = 2^j N / l.  Levels j and j + 1 make the same decisions -- and hold bit-identical buffers and snapshot
// rings -- until a direction falls between their thresholds (theta_j <= s2 < 2 theta_j: dumped at j, kept at j + 1);
// meta[.][3] counts those events since the sketch was cleared.  rep[s] = the highest level of the run of
// still-identical levels that starts at s: only representatives get a Gram matrix, an eigen-decomposition and a
// rotate product; the others read the representative's and apply their own threshold (results unchanged: identical
// inputs through the same deterministic kernels).  On the 8-blob stream of the benchmark 3 of 28 sketches are
// duplicates in steady state, 17 of 28 in the first window of a stream.  One thread per (lane, kind).
//
// DEAD MAIN sketches (round 2).  get() only ever reads the MAIN sketch of a level whose ring has lost no snapshot that
// is still inside the window (dropped + N <= now; the top level is the fallback), and at the first row of the next epoch
// MAIN is overwritten by AUX.  So once a MAIN sketch below the top level has lost a snapshot created in the CURRENT epoch
// (dropped > epoch start) it cannot be selected again before it is overwritten: nothing it computes from then on is
// observable.  Such sketches get rep = -1 and are skipped by the Gram / Jacobi / decide / rotate / scatter launches until
// the epoch ends -- 55 % of the MAIN rotations on the benchmark stream (levels 0-3 after three rotations, levels 4-9
// after ~32 of 78).  Exact: every get() returns what it would have returned; only `export_half(0)` of a dead level
// shows the frozen state.  (Never in the first epoch of a stream, where the AUX chain borrows MAIN's solves.)
__global__ void swfd_rep_kernel(const int* __restrict__ meta, int L, int nchains, int* __restrict__ rep, int twin,
                                const long long* __restrict__ dropped, long long epoch_start, int skip_dead,
                                long long* __restrict__ now_dev, long long now) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) *now_dev = now;  // (the rotation's time stamp: one launch less than a kernel of its own)
  if (c >= nchains) return;
  // first epoch of a stream: AUX is the twin of MAIN (both started empty) -> the AUX chain maps onto the MAIN chain
  const int base = ((twin && (c & 1)) ? c - 1 : c) * L;  // sketch index = lane * 2L + kind * L + level = chain * L + level
  if (twin && (c & 1)) {
    int r = base + L - 1;
    rep[c * L + L - 1] = r;
    for (int j = L - 2; j >= 0; --j) {
      if (meta[(base + j) * 4 + 3] != 0) r = base + j;
      rep[c * L + j] = r;
    }
    return;
  }
  int r = base + L - 1;
  rep[r] = r;
  for (int j = L - 2; j >= 0; --j) {
    if (meta[(base + j) * 4 + 3] != 0) r = base + j;
    rep[base + j] = r;
  }
  if (skip_dead && !twin && (c & 1) == 0)  // MAIN chain (sketch index = lane * 2L + kind * L + level, kind 0)
    for (int j = 0; j < L - 1; ++j)
      if (dropped[base + j] > epoch_start) rep[base + j] = -1;
}

// One workgroup per sketch: expiry, eigenvalue ordering, shrink, dump/keep plan, Wc.
// (evals, U): eigenvalues and eigenvectors, row pitch ldu; cols != 0: U holds the solver's raw columns
// lam_j u_j (column j contiguous) instead of the eigenvector matrix.
__global__ __launch_bounds__(1024) void swfd_decide_kernel(const double* __restrict__ evals,
                                                          const double* __restrict__ U, int ldu, int cols, int n2,
                                                          int ell, int cap,
                                                          int N, const long long* __restrict__ now_p,
                                                          const double* __restrict__ theta, int* __restrict__ meta,
                                                          long long* __restrict__ qt, long long* __restrict__ dropped,
                                                          int* __restrict__ plan, int* __restrict__ keep_src,
                                                          double* __restrict__ Wc, const int* __restrict__ rep) {
  __shared__ double lam[1024];
  __shared__ int order[1024];
  __shared__ double scale[512];  // sqrt(s2 / lam) of the top-l rows (0 = discarded)
  const int s = blockIdx.x, t = threadIdx.x;
  const long long now = *now_p;
  const int sr = rep ? rep[s] : s;  // whose eigen-decomposition this sketch uses (its own unless it is a duplicate)
  if (sr < 0) return;               // dead MAIN sketch (swfd_rep_kernel): frozen until the epoch ends
  const double* ev = evals + (long)sr * ldu;
  const double* Us = U + (long)sr * ldu * ldu;
  if (t < n2) lam[t] = ev[t];
  __syncthreads();
  if (t < n2) {
    const double mine = lam[t];
    int rank = 0;
    for (int j = 0; j < n2; ++j) {
      const double o = lam[j];
      rank += (o > mine) || (o == mine && j < t);
    }
    order[rank] = t;
  }
  __syncthreads();
  // ---- expiry, shrink, keep / dump plan: all threads (round 2; the same decisions, in the same order, as the serial
  // form: timestamps in the ring ascend from its head, the j-th dump of this rotation lands (head + count + j) mod cap
  // whatever is evicted meanwhile, and at most count old entries are evicted because a rotation dumps <= l = cap / 2) ----
  __shared__ int s_part[16];      // per-wave partial counts
  __shared__ int s_wcnt[3][4];    // per-wave totals of keep / dump / near among the l <= 256 directions (waves 0 .. 3)
  const int head0 = meta[s * 4 + 1], cnt0 = meta[s * 4 + 2];
  long long* q = qt + (long)s * cap;
  const int wv = t >> 6, ln = t & 63;
  // (a) expired snapshots: a prefix of the ring.  (Round 4: wave ballots / one exchange through LDS instead of workgroup-wide
  // scans -- the three prefix counts and this sum cost ~70 barriers of a 16-wave workgroup, most of the kernel's time.)
  int expired = 0;
  for (int k = t; k < cnt0; k += 1024) expired += (q[(head0 + k) % cap] + N <= now) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) expired += __shfl_xor(expired, o);
  if (ln == 0) s_part[wv] = expired;
  __syncthreads();
  int nexp = 0;
#pragma unroll
  for (int ww = 0; ww < 16; ++ww) nexp += s_part[ww];
  const int head1 = (head0 + nexp) % cap, cnt1 = cnt0 - nexp;
  // (b) per direction: shrunk energy, keep / dump
  const double l0 = lam[order[0]];
  const double delta = lam[order[ell - 1]] > 0.0 ? lam[order[ell - 1]] : 0.0;
  const double tol = 1e-10 * (l0 > 0.0 ? l0 : 0.0);
  const double th = theta[s];
  int kind = 0, near = 0;
  double sc = 0.0;
  if (t < ell) {
    const double l = lam[order[t]];
    double s2 = l - delta;
    s2 = s2 > 0.0 ? s2 : 0.0;
    if (s2 > tol) {
      sc = sqrt(s2 / l);
      if (cols) sc /= l;  // the raw column is lam u
      kind = (s2 >= th) ? 2 : 1;
      near = (kind == 2 && s2 < 2.0 * th) ? 1 : 0;  // dumped here but kept one level up: this sketch parts with the next level
    }
    scale[t] = sc;
  }
  // (c) positions: exclusive prefix counts of keeps and of dumps over the directions in order (directions live in the
  //     first l / 64 <= 4 waves: ballots inside a wave, the waves' totals through LDS)
  const unsigned long long mk = __ballot(kind == 1), md = __ballot(kind == 2), mn = __ballot(near != 0);
  const unsigned long long below = (ln == 0) ? 0ull : (~0ull >> (64 - ln));
  if (ln == 0 && wv < 4) {
    s_wcnt[0][wv] = __popcll(mk);
    s_wcnt[1][wv] = __popcll(md);
    s_wcnt[2][wv] = __popcll(mn);
  }
  __syncthreads();
  int kpos = __popcll(mk & below), dpos = __popcll(md & below);
  int nk = 0, nd = 0, nnear = 0;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    if (ww < wv) {
      kpos += s_wcnt[0][ww];
      dpos += s_wcnt[1][ww];
    }
    nk += s_wcnt[0][ww];
    nd += s_wcnt[1][ww];
    nnear += s_wcnt[2][ww];
  }
  const int nevict = (cnt1 + nd > cap) ? (cnt1 + nd - cap) : 0;  // <= cnt1
  // the newest evicted timestamp (read before the new stamps overwrite the ring)
  if (t == 0) {
    long long drop = dropped[s];
    if (nevict > 0) {
      const long long ts = q[(head1 + nevict - 1) % cap];
      drop = ts > drop ? ts : drop;
    }
    dropped[s] = drop;
  }
  __syncthreads();
  if (t < ell) {
    int pos = 0;
    if (kind == 1) {
      pos = kpos;
      keep_src[s * ell + kpos] = t;
    } else if (kind == 2) {
      pos = (head1 + cnt1 + dpos) % cap;
      q[pos] = now;
    }
    plan[(s * ell + t) * 2 + 0] = kind;
    plan[(s * ell + t) * 2 + 1] = pos;
  }
  if (t == 0) {
    meta[s * 4 + 0] = nk;
    meta[s * 4 + 1] = (head1 + nevict) % cap;
    meta[s * 4 + 2] = cnt1 + nd - nevict;
    meta[s * 4 + 3] += nnear;
  }
  __syncthreads();
  if (sr != s) return;  // the rotate product is taken from the representative
  double* W = Wc + (long)s * ell * n2;
  for (int e = t; e < ell * n2; e += 1024) {
    const int i = e / n2, a = e - i * n2;
    const double sc = scale[i];
    W[e] = (sc != 0.0) ? sc * (cols ? Us[(long)order[i] * ldu + a] : Us[(long)a * ldu + order[i]]) : 0.0;
  }
}

// rows of T -> kept positions of buf / snapshot ring; buffer rows >= nk are cleared
__global__ void swfd_scatter_kernel(const double* __restrict__ T, const int* __restrict__ plan,
                                    const int* __restrict__ keep_src, const int* __restrict__ meta,
                                    double* __restrict__ buf, double* __restrict__ queue, int n2, int ell, int cap,
                                    int d, const int* __restrict__ rep) {
  const int rho = blockIdx.x, s = blockIdx.y;
  if (rep && rep[s] < 0) return;  // dead MAIN sketch: frozen
  const double* Ts = T + (long)(rep ? rep[s] : s) * ell * d;
  if (rho < ell && plan[(s * ell + rho) * 2] == 2) {
    const int slot = plan[(s * ell + rho) * 2 + 1];
    double* dst = queue + ((long)s * cap + slot) * d;
    const double* src = Ts + (long)rho * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = src[c];
  }
  const int nk = meta[s * 4 + 0];
  double* dst = buf + ((long)s * n2 + rho) * d;
  if (rho < nk) {
    const double* src = Ts + (long)keep_src[s * ell + rho] * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = src[c];
  } else {
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = 0.0;
  }
}

static int swfd_rotate_all(Swfd* h, hipStream_t st) {
  const int S = h->S, n2 = h->n2, d = h->d, ell = h->ell;
  int rc;
  if (h->rep)
    hipLaunchKernelGGL(swfd_rep_kernel, dim3(cdiv(2 * h->lanes, 64)), dim3(64), 0, st, h->meta, h->L, 2 * h->lanes, h->rep,
                       h->twin ? 1 : 0, h->dropped, (long long)(((h->i - 1) / h->N) * h->N), h->skip_dead, h->now_dev,
                       (long long)h->i);
  else
    hipLaunchKernelGGL(swfd_set_now_kernel, dim3(1), dim3(1), 0, st, h->now_dev, (long long)h->i);
  if (h->gram_split > 1) {
    const int kchunk = ((d + h->gram_split - 1) / h->gram_split + 15) / 16 * 16;
    if ((rc = gemm_f64_batched_splitk(true, true, h->buf, d, (long)n2 * d, h->buf, d, (long)n2 * d, h->gpart, n2, n2, d, S, kchunk,
                                      h->gram_split, st, h->rep)))
      return rc;
    if ((rc = gemm_batched_splitk_reduce(h->gpart, h->gram_split, (long)n2 * n2, eig_plan_input(h->eig), S, h->rep, st))) return rc;
  } else if ((rc = gemm_f64(true, true, h->buf, d, (long)n2 * d, h->buf, d, (long)n2 * d, eig_plan_input(h->eig), n2,
                            (long)n2 * n2, n2, n2, d, S, 1.0, st, h->rep)))
    return rc;
  const double *ecols = nullptr, *elam = nullptr;
  int eld = 0;
  const bool raw = eig_plan_columns(h->eig, &ecols, &elam, &eld);
  if ((rc = eig_plan_run_inplace(h->eig, raw ? nullptr : h->evals, raw ? nullptr : h->U, st, true))) return rc;
  hipLaunchKernelGGL(swfd_decide_kernel, dim3(S), dim3(1024), 0, st, raw ? elam : h->evals, raw ? ecols : h->U,
                     raw ? eld : n2, raw ? 1 : 0, n2, ell, h->cap, h->N,
                     h->now_dev, h->theta, h->meta, h->qt, h->dropped, h->plan, h->keep_src, h->Wc, h->rep);
  if ((rc = gemm_f64(true, false, h->Wc, n2, (long)ell * n2, h->buf, d, (long)n2 * d, h->T, d, (long)ell * d, ell, d,
                     n2, S, 1.0, st, h->rep)))
    return rc;
  hipLaunchKernelGGL(swfd_scatter_kernel, dim3(n2, S), dim3(256), 0, st, h->T, h->plan, h->keep_src, h->meta, h->buf,
                     h->queue, n2, ell, h->cap, d, h->rep);
  MUSED_LAUNCH_CHECK();
  h->pend = 0;
  return MUSED_OK;
}

static int swfd_restart(Swfd* h, hipStream_t st) {
  const long be = (long)h->n2 * h->d, qe = (long)h->cap * h->d;
  const long m = be > qe ? be : qe;
  hipLaunchKernelGGL(swfd_restart_kernel, dim3(cdiv(m, 256), h->L * h->lanes), dim3(256), 0, st, h->buf, h->queue, h->qt,
                     h->meta, h->dropped, h->L, be, qe, h->cap);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// ---- pre-rotation of the input blocks ----------------------------------------------------------------------
// A rotation sees buf = [kept rows K (mutually orthogonal); the l rows P that arrived since the last one].  Replacing
// P by V^T P with an orthogonal V changes nothing in the sketch (P^T P is what enters), and if V (nearly) diagonalises
// P P^T both diagonal blocks of the Gram matrix are diagonal: the Jacobi of the rotation then needs about two sweeps
// less (10.5 -> 8.5 on the benchmark stream).  P is the same for all 2L sketches of a lane and known as soon as the
// rows are: every complete block of an append call is rotated beforehand, (lanes x blocks) at a time.
// V comes from a few fixed sweeps of the one-sided Jacobi on the stacked matrix [P P^T ; I] (zero-padded to order 2l):
// the rotations that orthogonalise its columns turn the lower half into V itself, orthogonal to rounding whether or
// not the sweeps have converged, rank-deficient blocks included.
constexpr int SWFD_PRE_MAX = 96;  // blocks per batch (the table travels as a kernel argument)
struct PreBlocks {
  int row0[SWFD_PRE_MAX];  // first row of the block in the rows of this append call
  int rows[SWFD_PRE_MAX];  // its row count (<= l; shorter at the end of an epoch: zero padded)
};

template <typename T>
__global__ void swfd_pre_load_kernel(const T* __restrict__ X, long ldx, long lane_stride, int lanes, int ell, int d,
                                     PreBlocks pb, double* __restrict__ out) {
  const int row = blockIdx.x, z = blockIdx.y;  // z = block * lanes + lane
  const int blk = z / lanes, lane = z - blk * lanes;
  double* dst = out + ((long)z * ell + row) * d;
  if (row < pb.rows[blk]) {
    const T* src = X + (long)lane * lane_stride + (long)(pb.row0[blk] + row) * ldx;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = row_elem<T>(src, c);
  } else {
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = 0.0;
  }
}

// Gc (order 2l, column-major): column j < l = [gram[:, j] ; e_j], columns >= l zero
__global__ void swfd_pre_pack_kernel(const double* __restrict__ gram, int ell, double* __restrict__ Gc) {
  const int n2 = 2 * ell;
  const long z = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)n2 * n2) return;
  const int col = (int)(e / n2), row = (int)(e - (long)col * n2);
  double v = 0.0;
  if (col < ell) v = (row < ell) ? gram[z * ell * ell + (long)col * ell + row] : ((row - ell == col) ? 1.0 : 0.0);
  Gc[z * n2 * n2 + e] = v;
}

template <typename T>
static int swfd_prerotate(Swfd* h, const T* X, long ldx, long lane_stride, const PreBlocks& pb, int nblk, hipStream_t st) {
  const int ell = h->ell, d = h->d, B = h->lanes, n2 = h->n2;
  int rc;
  const int Z = nblk * B;
  hipLaunchKernelGGL(swfd_pre_load_kernel<T>, dim3(ell, Z), dim3(256), 0, st, X, ldx, lane_stride, B, ell, d, pb,
                     h->pre_in);
  if ((rc = gemm_f64(true, true, h->pre_in, d, (long)ell * d, h->pre_in, d, (long)ell * d, h->pre_gram, ell,
                     (long)ell * ell, ell, ell, d, Z, 1.0, st)))
    return rc;
  hipLaunchKernelGGL(swfd_pre_pack_kernel, dim3(cdiv((long)n2 * n2, 256), Z), dim3(256), 0, st, h->pre_gram, ell,
                     eig_plan_input(h->eigp));
  // matrices of the plan beyond Z keep whatever they held (their results are not read)
  if ((rc = eig_plan_run_inplace(h->eigp, nullptr, nullptr, st, true))) return rc;
  const double *cols = nullptr, *lam = nullptr;
  int ld = 0;
  if (!eig_plan_columns(h->eigp, &cols, &lam, &ld)) return MUSED_ERR_STATE;
  // P' = V^T P, V = rows l .. 2l-1 of the first l columns of the solver's working copy: A[j][i] = cols[j * ld + l + i]
  if ((rc = gemm_f64(true, false, cols + ell, ld, (long)ld * ld, h->pre_in, d, (long)ell * d, h->pre_out, d, (long)ell * d,
                     ell, d, ell, Z, 1.0, st)))
    return rc;
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

template <typename T>
static int swfd_append_t(Swfd* h, const T* X, long ldx, long lane_stride, long m, hipStream_t st) {
  long r = 0;
  int rc;
  // blocks of this call already rotated into pre_out: [pre_first, pre_first + pre_n), in call-row coordinates
  PreBlocks pb;
  int pre_n = 0, pre_next = 0;
  while (r < m) {
    if (h->i > 0 && h->i % h->N == 0 && h->restart_mark != h->i) {
      // first row of a new epoch (pend == 0 here: the epoch-end rotation has run)
      if ((rc = swfd_restart(h, st))) return rc;
      h->restart_mark = h->i;
      h->twin = false;  // MAIN now carries the previous epoch, AUX starts empty
    }
    const long in_epoch = h->i % h->N;
    long until = h->ell - (in_epoch % h->ell);
    if (h->N - in_epoch < until) until = h->N - in_epoch;
    const long take = (m - r) < until ? (m - r) : until;
    bool from_pre = false;
    if (h->pre_chunk > 0 && h->pend == 0 && take == until) {
      // a complete block: rotate it (and the complete blocks that follow in this call, a batch at a time) beforehand
      if (pre_next >= pre_n) {
        pre_n = pre_next = 0;
        long rr = r, ii = h->i;
        while (pre_n < h->pre_chunk && rr < m) {
          const long ie = ii % h->N;
          long u = h->ell - (ie % h->ell);
          if (h->N - ie < u) u = h->N - ie;
          if (m - rr < u) break;  // the call ends inside this block
          pb.row0[pre_n] = (int)rr;
          pb.rows[pre_n] = (int)u;
          ++pre_n;
          rr += u;
          ii += u;
        }
        if ((rc = swfd_prerotate<T>(h, X, ldx, lane_stride, pb, pre_n, st))) return rc;
      }
      from_pre = true;
    }
    if (from_pre) {
      const double* src = h->pre_out + (long)pre_next * h->lanes * h->ell * h->d;
      hipLaunchKernelGGL(swfd_append_kernel<double>, dim3((int)take, h->S), dim3(256), 0, st, src, (long)h->d,
                         (long)h->ell * h->d, 2 * h->L, (int)take, h->d, h->n2, h->pend, h->meta, h->buf);
      ++pre_next;
    } else {
      hipLaunchKernelGGL(swfd_append_kernel<T>, dim3((int)take, h->S), dim3(256), 0, st, X + r * ldx, ldx, lane_stride,
                         2 * h->L, (int)take, h->d, h->n2, h->pend, h->meta, h->buf);
    }
    MUSED_LAUNCH_CHECK();
    h->pend += (int)take;
    h->i += take;
    r += take;
    if (take == until) {
      if ((rc = swfd_rotate_all(h, st))) return rc;
    }
  }
  return MUSED_OK;
}

// ---- query ---------------------------------------------------------------------------
__global__ void swfd_select_kernel(const int* __restrict__ meta, const long long* __restrict__ qt,
                                   const long long* __restrict__ dropped, int L, int cap, int N, long long now,
                                   int* __restrict__ qsel_all) {
  if (threadIdx.x != 0) return;
  const int lane = blockIdx.x;
  const int base = lane * 2 * L;  // MAIN sketches of this lane are base .. base + L - 1
  int* qsel = qsel_all + (long)lane * (4 + cap);
  int lvl = L - 1;
  for (int j = 0; j < L; ++j) {
    const long long dr = dropped[base + j];
    if (dr == 0 || dr + N <= now) { lvl = j; break; }
  }
  const int sk = base + lvl;
  const int head = meta[sk * 4 + 1], cnt = meta[sk * 4 + 2];
  int ns = 0;
  for (int q = 0; q < cnt; ++q) {
    const int slot = (head + q) % cap;
    if (qt[(long)sk * cap + slot] + N > now) qsel[4 + ns++] = slot;
  }
  qsel[0] = lvl;
  qsel[1] = ns;
  qsel[2] = meta[sk * 4 + 0];
  qsel[3] = sk;
}

__global__ void swfd_stack_kernel(const int* __restrict__ qsel_all, const double* __restrict__ buf,
                                  const double* __restrict__ queue, int n2, int n4, int cap, int d, int pend,
                                  double* __restrict__ stack) {
  const int rho = blockIdx.x, lane = blockIdx.y;
  const int* qsel = qsel_all + (long)lane * (4 + cap);
  const int ns = qsel[1], nk = qsel[2], sk = qsel[3];
  const double* src = nullptr;
  if (rho < ns) src = queue + ((long)sk * cap + qsel[4 + rho]) * d;
  else if (rho < ns + nk + pend) src = buf + ((long)sk * n2 + (rho - ns)) * d;
  double* dst = stack + ((long)lane * n4 + rho) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = src ? src[c] : 0.0;
}

__global__ __launch_bounds__(1024) void swfd_decide_query_kernel(const double* __restrict__ evals_all,
                                                                const double* __restrict__ U_all, int ldu, int cols,
                                                                int n4, int ell,
                                                                int cap, const int* __restrict__ qsel_all,
                                                                double* __restrict__ Wq_all,
                                                                double* __restrict__ qinfo_all) {
  __shared__ double lam[1024];
  __shared__ int order[1024];
  __shared__ double scale[512];
  const int t = threadIdx.x, lane = blockIdx.x;
  const double* evals = evals_all + (long)lane * ldu;
  const double* U = U_all + (long)lane * ldu * ldu;
  const int* qsel = qsel_all + (long)lane * (4 + cap);
  double* Wq = Wq_all + (long)lane * ell * n4;
  double* qinfo = qinfo_all + (long)lane * 2;
  if (t < n4) lam[t] = evals[t];
  __syncthreads();
  if (t < n4) {
    const double mine = lam[t];
    int rank = 0;
    for (int j = 0; j < n4; ++j) {
      const double o = lam[j];
      rank += (o > mine) || (o == mine && j < t);
    }
    order[rank] = t;
  }
  __syncthreads();
  const double l0 = lam[order[0]];
  const double delta = lam[order[ell - 1]] > 0.0 ? lam[order[ell - 1]] : 0.0;
  const double tol = 1e-10 * (l0 > 0.0 ? l0 : 0.0);
  if (t < ell) {
    const double l = lam[order[t]];
    double s2 = l - delta;
    s2 = s2 > 0.0 ? s2 : 0.0;
    scale[t] = (s2 > tol) ? (cols ? sqrt(s2 / l) / l : sqrt(s2 / l)) : 0.0;
  }
  if (t == 0) {
    qinfo[0] = (double)qsel[0];
    qinfo[1] = delta;
  }
  __syncthreads();
  for (int e = t; e < ell * n4; e += 1024) {
    const int i = e / n4, a = e - i * n4;
    const double sc = scale[i];
    Wq[e] = (sc != 0.0) ? sc * (cols ? U[(long)order[i] * ldu + a] : U[(long)a * ldu + order[i]]) : 0.0;
  }
}

// per output row: largest-magnitude entry made positive (first index on ties), sigma = row norm
__global__ __launch_bounds__(256) void swfd_finish_rows_kernel(double* __restrict__ B, int d, double* __restrict__ sig) {
  __shared__ double sv[4];
  __shared__ int si[4];
  __shared__ double ss[4];
  __shared__ double s_sign;
  const long rowid = (long)blockIdx.y * gridDim.x + blockIdx.x;
  double* row = B + rowid * d;
  double best = -1.0, sq = 0.0;
  int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < d; c += 256) {
    const double v = row[c];
    const double a = fabs(v);
    sq += v * v;
    if (a > best || (a == best && c < bi)) { best = a; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    sq += __shfl_xor(sq, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; ss[threadIdx.x >> 6] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    s_sign = (row[bi] < 0.0) ? -1.0 : 1.0;
    sig[rowid] = sqrt(ss[0] + ss[1] + ss[2] + ss[3]);
  }
  __syncthreads();
  if (s_sign < 0.0)
    for (int c = threadIdx.x; c < d; c += 256) row[c] = -row[c];
}

static int swfd_query(Swfd* h, double* outB, double* outSigma, double* outInfo, hipStream_t st) {
  // stacked rows = in-window snapshots (<= 2l) + kept rows (<= l - 1) + pending rows: a query on a rotation boundary
  // (every window boundary is one) has nothing pending and fits order 3l -- 0.42 of the eigenwork of order 4l
  const bool small = (h->pend == 0) && h->eigq3;
  EigPlan* const eq = small ? h->eigq3 : h->eigq;
  const int n4 = small ? h->n3 : h->n4, d = h->d, ell = h->ell, B = h->lanes;
  int rc;
  hipLaunchKernelGGL(swfd_select_kernel, dim3(B), dim3(64), 0, st, h->meta, h->qt, h->dropped, h->L, h->cap, h->N,
                     (long long)h->i, h->qsel);
  hipLaunchKernelGGL(swfd_stack_kernel, dim3(n4, B), dim3(256), 0, st, h->qsel, h->buf, h->queue, h->n2, n4, h->cap, d,
                     h->pend, h->stack);
  if ((rc = gemm_f64(true, true, h->stack, d, (long)n4 * d, h->stack, d, (long)n4 * d, eig_plan_input(eq), n4,
                     (long)n4 * n4, n4, n4, d, B, 1.0, st)))
    return rc;
  const double *ecols = nullptr, *elam = nullptr;
  int eld = 0;
  const bool raw = eig_plan_columns(eq, &ecols, &elam, &eld);
  if ((rc = eig_plan_run_inplace(eq, raw ? nullptr : h->evals_q, raw ? nullptr : h->Uq, st, true))) return rc;
  hipLaunchKernelGGL(swfd_decide_query_kernel, dim3(B), dim3(1024), 0, st, raw ? elam : h->evals_q, raw ? ecols : h->Uq,
                     raw ? eld : n4, raw ? 1 : 0, n4, ell, h->cap, h->qsel,
                     h->Wq, h->qinfo);
  if ((rc = gemm_f64(true, false, h->Wq, n4, (long)ell * n4, h->stack, d, (long)n4 * d, h->Bout, d, (long)ell * d, ell,
                     d, n4, B, 1.0, st)))
    return rc;
  hipLaunchKernelGGL(swfd_finish_rows_kernel, dim3(ell, B), dim3(256), 0, st, h->Bout, d, h->sig_out);
  MUSED_LAUNCH_CHECK();
  MUSED_CHECK_HIP(hipMemcpyAsync(outB, h->Bout, sizeof(double) * (size_t)B * ell * d, hipMemcpyDeviceToDevice, st));
  if (outSigma)
    MUSED_CHECK_HIP(hipMemcpyAsync(outSigma, h->sig_out, sizeof(double) * (size_t)B * ell, hipMemcpyDeviceToDevice, st));
  if (outInfo) MUSED_CHECK_HIP(hipMemcpyAsync(outInfo, h->qinfo, sizeof(double) * 2 * B, hipMemcpyDeviceToDevice, st));
  return MUSED_OK;
}

static size_t swfd_half_bytes(const Swfd* h) {
  const size_t L = h->L;
  return L * ((size_t)h->n2 * h->d * 8 + (size_t)h->cap * h->d * 8 + (size_t)h->cap * 8 + 4 * 4 + 8);
}

}  // namespace mused

using namespace mused;

extern "C" {

// Replaces SeqBasedSWFD.__init__ (call site main.py:62: N=window_size, R=max row norm^2,
// d=row length, sketch_dim=l).  sweeps: Jacobi sweeps per rotation (0 -> default).
int mused_swfd_create_lanes(long N, double R, int d, int ell, int sweeps, int lanes, void** out);
int mused_swfd_destroy(void* handle);

int mused_swfd_create(long N, double R, int d, int ell, int sweeps, void** out) {
  return mused_swfd_create_lanes(N, R, d, ell, sweeps, 1, out);
}

// `lanes` independent sketch sets (e.g. the windows of `lanes` contiguous blocks of the stream) advanced
// in lockstep by the same launches: every append feeds the same number of rows to every lane.
static int swfd_create_impl(Swfd* h, long N, double R, int d, int ell, int sweeps, int lanes);

int mused_swfd_create_lanes(long N, double R, int d, int ell, int sweeps, int lanes, void** out) {
  MUSED_REQUIRE(out && N >= 1 && d >= 1 && ell >= 1 && ell <= 256, "mused_swfd_create: need N, d >= 1 and 1 <= sketch_dim <= 256");
  MUSED_REQUIRE(lanes >= 1 && lanes <= 64, "mused_swfd_create_lanes: 1 <= lanes <= 64");
  MUSED_REQUIRE(N < (1l << 31), "mused_swfd_create: N too large");
  CaptureLock resource_guard(capture_mutex());  // allocations + the plans' captures: not beside another thread's capture
  Swfd* h = new Swfd();
  memset(h, 0, sizeof(*h));
  const int rc = swfd_create_impl(h, N, R, d, ell, sweeps, lanes);
  if (rc) {  // a later allocation failed: release what the earlier ones took (the error message is kept)
    (void)mused_swfd_destroy(h);
    return rc;
  }
  *out = h;
  return MUSED_OK;
}

static int swfd_create_impl(Swfd* h, long N, double R, int d, int ell, int sweeps, int lanes) {
  h->N = (int)N; h->R = R; h->d = d; h->ell = ell;
  double r1 = R > 1.0 ? R : 1.0;
  int lg = 0;
  while ((double)(1ull << lg) < r1 && lg < 62) ++lg;  // ceil(log2(R))
  h->L = lg + 1;
  h->lanes = lanes;
  h->S = 2 * h->L * lanes;
  h->cap = 2 * ell;
  h->n2 = 2 * ell;
  h->n4 = 4 * ell;
  h->n3 = (3 * ell + 1) & ~1;
  // cap of the adaptive sweep count: full-rank buffers stop after 10-11 sweeps, dense rank-deficient Gram
  // matrices (d < 2l, or linearly dependent rows) need up to 20 to push the null-space columns below the drop tolerance
  h->sweeps = sweeps > 0 ? sweeps : 24;
  h->restart_mark = -1;
  h->twin = true;  // both halves empty
  const size_t S = h->S, n2 = h->n2, n4 = h->n4, cap = h->cap, dd = d, l = ell;
#define ALLOC(p, bytes) MUSED_CHECK_HIP(hipMalloc((void**)&(p), (bytes)))
#define ZALLOC(p, bytes) do { ALLOC(p, bytes); MUSED_CHECK_HIP(hipMemset((p), 0, (bytes))); } while (0)
  ZALLOC(h->buf, 8 * S * n2 * dd);
  ZALLOC(h->queue, 8 * S * cap * dd);
  ZALLOC(h->qt, 8 * S * cap);
  ZALLOC(h->meta, 4 * S * 4);
  {
    const char* dd = getenv("MUSED_SWFD_DEDUPE");
    if (!(dd && dd[0] == '0')) ALLOC(h->rep, 4 * S);
    const char* sd = getenv("MUSED_SWFD_SKIP_DEAD");
    h->skip_dead = (sd && sd[0] == '0') ? 0 : 1;
  }
  ZALLOC(h->dropped, 8 * S);
  ZALLOC(h->status, 4);
  MUSED_CHECK_HIP(hipHostMalloc((void**)&h->status_host, sizeof(int), hipHostMallocDefault));
  ALLOC(h->theta, 8 * S);
  ALLOC(h->T, 8 * S * l * dd); ALLOC(h->Wc, 8 * S * l * n2); ALLOC(h->evals, 8 * S * n2); ALLOC(h->U, 8 * S * n2 * n2);
  ALLOC(h->plan, 4 * S * l * 2); ALLOC(h->keep_src, 4 * S * l); ALLOC(h->now_dev, 8);
  {
    // rows of 8,192 entries and more (the SWFDMC wiring: d = window size): eight K-slices per tile (measured at d = 10,000, 12 lanes:
    // 181 -> 186 / 190 k rows/s with 4 / 8 slices, 185 k with 16).  Decided by d alone -- never by the batch: lock-step lanes and
    // single sketches must agree bit for bit.  MUSED_SWFD_GRAM_SPLIT=n overrides (1: off).
    const char* gs = getenv("MUSED_SWFD_GRAM_SPLIT");
    h->gram_split = gs ? atoi(gs) : (d >= 8192 ? 8 : 1);
    if (h->gram_split < 1 || h->gram_split > 16) h->gram_split = 1;
    if (h->gram_split > 1) ALLOC(h->gpart, 8 * S * (size_t)h->gram_split * n2 * n2);
  }
  const size_t Bn = lanes;
  ALLOC(h->stack, 8 * Bn * n4 * dd); ALLOC(h->evals_q, 8 * Bn * n4); ALLOC(h->Uq, 8 * Bn * n4 * n4);
  ALLOC(h->Wq, 8 * Bn * l * n4); ALLOC(h->Bout, 8 * Bn * l * dd); ALLOC(h->sig_out, 8 * Bn * l);
  ALLOC(h->qinfo, 8 * 2 * Bn); ALLOC(h->qsel, 4 * Bn * (4 + cap));
#undef ZALLOC
#undef ALLOC
  std::vector<double> th(S);
  for (int lane = 0; lane < lanes; ++lane)
    for (int kind = 0; kind < 2; ++kind)
      for (int j = 0; j < h->L; ++j) th[sk_index(h, lane, j, kind)] = ldexp((double)h->N / ell, j);
  MUSED_CHECK_HIP(hipMemcpy(h->theta, th.data(), 8 * S, hipMemcpyHostToDevice));
  int rc;
  if ((rc = gemm_f64_prepare_all())) return rc;
  if ((rc = eig_plan_create(h->n2, h->S, h->sweeps, true, &h->eig, h->rep, EIG_PLAN_TOP_HALF, h->status))) return rc;
  // the query reads the l largest pairs of its stacked Gram (order 4 l, or 3 l on a rotation boundary) with the same shrink
  // rule as a rotation: orders the direct solver covers go to it
  if ((rc = eig_plan_create(h->n4, lanes, h->sweeps + 2, true, &h->eigq, nullptr, EIG_PLAN_TOP_FD, h->status, ell))) return rc;
  if (h->n3 < h->n4 &&
      (rc = eig_plan_create(h->n3, lanes, h->sweeps + 2, true, &h->eigq3, nullptr, EIG_PLAN_TOP_FD, h->status, ell)))
    return rc;
  {
    // input-block pre-rotation (swfd_prerotate): batches of `pre_chunk` blocks x lanes, workspace <= ~256 MB per array
    const char* pr = getenv("MUSED_SWFD_PREROT");
    const double *c0 = nullptr, *l0 = nullptr;
    int ld0 = 0;
    // (the pre-rotation only saves Jacobi sweeps: with the direct solver it is off unless MUSED_SWFD_PREROT=1 asks for it)
    const bool want_pre = pr ? (pr[0] != '0') : !eig_plan_direct_solver(h->eig);
    if (want_pre && eig_plan_columns(h->eig, &c0, &l0, &ld0)) {
      const size_t per_block = (size_t)lanes * ell * dd * 8;
      size_t c = ((size_t)256 << 20) / per_block;
      c = c < 1 ? 1 : (c > (size_t)SWFD_PRE_MAX ? (size_t)SWFD_PRE_MAX : c);
      h->pre_chunk = (int)c;
      MUSED_CHECK_HIP(hipMalloc((void**)&h->pre_in, c * per_block));
      MUSED_CHECK_HIP(hipMalloc((void**)&h->pre_out, c * per_block));
      MUSED_CHECK_HIP(hipMalloc((void**)&h->pre_gram, 8 * c * lanes * l * l));
      const char* ps = getenv("MUSED_SWFD_PREROT_SWEEPS");
      const int psw = ps ? atoi(ps) : 5;
      if ((rc = eig_plan_create(h->n2, (int)c * lanes, psw > 0 ? psw : 5, true, &h->eigp, nullptr,
                                EIG_PLAN_FIXED_SWEEPS | EIG_PLAN_NO_SORT, h->status)))
        return rc;
    }
  }
  return MUSED_OK;
}

int mused_swfd_destroy(void* handle) {
  Swfd* h = (Swfd*)handle;
  if (!h) return MUSED_OK;
  CaptureLock resource_guard(capture_mutex());
  eig_plan_destroy(h->eig);
  eig_plan_destroy(h->eigq);
  if (h->eigq3) eig_plan_destroy(h->eigq3);
  if (h->eigp) eig_plan_destroy(h->eigp);
  void* bufs[] = {h->buf, h->queue, h->qt, h->meta, h->dropped, h->theta, h->T, h->Wc, h->evals, h->U, h->plan,
                  h->keep_src, h->now_dev, h->stack, h->evals_q, h->Uq, h->Wq, h->Bout, h->sig_out, h->qinfo, h->qsel,
                  h->rep, h->pre_in, h->pre_out, h->pre_gram, h->status, h->gpart};
  for (void* b : bufs) (void)hipFree(b);
  if (h->status_host) (void)hipHostFree(h->status_host);
  delete h;
  return MUSED_OK;
}

int mused_swfd_levels(void* handle) { return handle ? ((Swfd*)handle)->L : -1; }

// Live timing of the rotation eigensolver (the dominant kernel of the path): enable, run appends, then read
// (BLOCKING) the summed duration of the Jacobi sweep graphs, the number of osj_round_kernel launches in
// them and the bytes one launch streams (all S matrices of order 2l read once and written once).
int mused_swfd_profile(void* handle, int on) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h, "mused_swfd_profile: null handle");
  return eig_plan_profile(h->eig, on != 0);
}

int mused_swfd_profile_read(void* handle, double* total_ms, long* launches, double* bytes_per_launch) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && total_ms && launches && bytes_per_launch, "mused_swfd_profile_read: null pointer");
  return eig_plan_profile_read(h->eig, total_ms, launches, bytes_per_launch);
}

// As mused_swfd_profile_read for sketches whose rotations run the direct eigensolver (csrc/trd.hip; order 2 l = 256):
// summed ms of the direct-solver launches (trd_a .. trd_d, five kernels each) timed since mused_swfd_profile(handle, 1), their
// number, the number of matrices they solved, and (tridiag_ms, may be NULL) the part of that time up to the end of trd_a_kernel.  *direct = 0: this sketch's rotations run the Jacobi (use mused_swfd_profile_read).  BLOCKING.
extern "C" int mused_swfd_profile_read_direct(void* handle, double* total_ms, long* launches, double* matrices_solved,
                                              int* direct, double* tridiag_ms) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && total_ms && launches && matrices_solved && direct, "mused_swfd_profile_read_direct: null pointer");
  *direct = eig_plan_direct_solver(h->eig) ? 1 : 0;
  return eig_plan_profile_read_direct(h->eig, total_ms, launches, matrices_solved, tridiag_ms);
}

// Replaces the per-row SeqBasedSWFD.fit(row) loop (main.py:65-67): appends n_rows rows of
// length d (device pointer, row pitch ld elements, dtype MUSED_F32 / F64 / I64 -- the fused matrix
// is int64 for >= 2 modalities, matrix_operations.py:138).  Any split of the stream into calls
// gives the same sketch.
int mused_swfd_append_lanes(void* handle, const void* rows, int dtype, long n_rows, long ld, long lane_stride,
                            void* stream);

int mused_swfd_append(void* handle, const void* rows, int dtype, long n_rows, long ld, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && h->lanes == 1, "mused_swfd_append: this handle has %d lanes, use mused_swfd_append_lanes", h ? h->lanes : 0);
  return mused_swfd_append_lanes(handle, rows, dtype, n_rows, ld, 0, stream);
}

// n_rows rows for EVERY lane: lane b reads rows + b * lane_stride (elements), row pitch ld.
int mused_swfd_append_lanes(void* handle, const void* rows, int dtype, long n_rows, long ld, long lane_stride,
                            void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && (rows || n_rows == 0) && n_rows >= 0, "mused_swfd_append: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MUSED_BITS) {  // rows of a 0/1 matrix as bitmask rows, pitch in 64-bit words
    MUSED_REQUIRE(ld >= (h->d + 63) / 64, "mused_swfd_append: bit rows need a pitch of at least ceil(d / 64) words");
    return swfd_append_t<BitWord>(h, (const BitWord*)rows, ld, lane_stride, n_rows, st);
  }
  MUSED_REQUIRE(ld >= h->d, "mused_swfd_append: row pitch smaller than d");
  if (dtype == MUSED_F32) return swfd_append_t<float>(h, (const float*)rows, ld, lane_stride, n_rows, st);
  if (dtype == MUSED_F64) return swfd_append_t<double>(h, (const double*)rows, ld, lane_stride, n_rows, st);
  if (dtype == MUSED_I64) return swfd_append_t<long long>(h, (const long long*)rows, ld, lane_stride, n_rows, st);
  set_error("mused_swfd_append: unsupported dtype %d", dtype);
  return MUSED_ERR_UNSUPPORTED;
}

int mused_swfd_lanes(void* handle) { return handle ? ((Swfd*)handle)->lanes : -1; }

// Diagnostic (not part of the declared ABI): copies rep[S] and meta[S][4] of the last rotation to host arrays.
int mused_swfd_debug_state(void* handle, int* rep_out, int* meta_out) {
  Swfd* h = (Swfd*)handle;
  if (!h) return MUSED_ERR_ARG;
  MUSED_CHECK_HIP(hipDeviceSynchronize());
  if (h->rep && rep_out) MUSED_CHECK_HIP(hipMemcpy(rep_out, h->rep, 4 * (size_t)h->S, hipMemcpyDeviceToHost));
  if (meta_out) MUSED_CHECK_HIP(hipMemcpy(meta_out, h->meta, 16 * (size_t)h->S, hipMemcpyDeviceToHost));
  return h->rep ? 1 : 0;
}

// Replaces SeqBasedSWFD.get() (main.py:70): out_sketch (l x d fp64), out_sigma (l, row norms of the
// sketch = its singular values), out_info = {level used, delta of the final shrink}.  Device pointers.
int mused_swfd_query(void* handle, double* out_sketch, double* out_sigma, double* out_info, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && out_sketch, "mused_swfd_query: null pointer");
  return swfd_query(h, out_sketch, out_sigma, out_info, (hipStream_t)stream);
}

// BLOCKING status read (synchronises `stream`): *status_out != 0 -> an eigensolve of this sketch gave up (bit 0: timeout of
// the persistent work-queue solver) and everything the sketch has returned since is INVALID.  Sticky until destroy.
int mused_swfd_status(void* handle, int* status_out, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && status_out, "mused_swfd_status: null pointer");
  // an asynchronous copy on the caller's stream into pinned memory + a wait for that stream only: no legacy-stream
  // blocking copy (device-wide synchronisation next to another thread's stream capture is what kills the capture)
  MUSED_CHECK_HIP(hipMemcpyAsync(h->status_host, h->status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MUSED_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  *status_out = *h->status_host;
  return MUSED_OK;
}

int mused_swfd_counters(void* handle, long* rows_seen, int* pending) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && rows_seen && pending, "mused_swfd_counters: null pointer");
  *rows_seen = h->i;
  *pending = h->pend;
  return MUSED_OK;
}

// ---- state exchange between ranks (windows sharded over GPUs) --------------------
// One "half" = the L sketches of one kind (0 = MAIN, 1 = AUX) packed as
// [bufs | queues | timestamps | meta | dropped].
long mused_swfd_half_bytes(void* handle) { return handle ? (long)swfd_half_bytes((Swfd*)handle) : -1; }

static int half_copy(Swfd* h, int kind, char* blob, bool to_blob, hipStream_t st) {
  const size_t L = h->L;
  const size_t o = (size_t)kind * L;
  struct Part { char* dev; size_t bytes; };
  Part parts[5] = {
      {(char*)(h->buf + o * h->n2 * h->d), L * h->n2 * h->d * 8},
      {(char*)(h->queue + o * h->cap * h->d), L * (size_t)h->cap * h->d * 8},
      {(char*)(h->qt + o * h->cap), L * (size_t)h->cap * 8},
      {(char*)(h->meta + o * 4), L * 16},
      {(char*)(h->dropped + o), L * 8},
  };
  size_t off = 0;
  for (const Part& p : parts) {
    if (to_blob) MUSED_CHECK_HIP(hipMemcpyAsync(blob + off, p.dev, p.bytes, hipMemcpyDeviceToDevice, st));
    else MUSED_CHECK_HIP(hipMemcpyAsync(p.dev, blob + off, p.bytes, hipMemcpyDeviceToDevice, st));
    off += p.bytes;
  }
  return MUSED_OK;
}

int mused_swfd_export_half(void* handle, int kind, void* dst, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && dst && (kind == 0 || kind == 1), "mused_swfd_export_half: bad arguments");
  MUSED_REQUIRE(h->lanes == 1, "mused_swfd_export_half: single-lane handles only");
  return half_copy(h, kind, (char*)dst, true, (hipStream_t)stream);
}

int mused_swfd_import_half(void* handle, int kind, const void* src, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && src && (kind == 0 || kind == 1), "mused_swfd_import_half: bad arguments");
  MUSED_REQUIRE(h->lanes == 1, "mused_swfd_import_half: single-lane handles only");
  h->twin = false;
  return half_copy(h, kind, (char*)src, false, (hipStream_t)stream);
}

// Start epoch e = rows_seen / N on this rank with the row counter set to `rows_seen` (a multiple
// of N): AUX is cleared; MAIN is taken from `main_half` (the AUX half exported by the rank that
// processed the previous window) or, if null, cleared as well (stream start).
int mused_swfd_begin_epoch(void* handle, long rows_seen, const void* main_half, void* stream) {
  Swfd* h = (Swfd*)handle;
  MUSED_REQUIRE(h && rows_seen >= 0 && rows_seen % h->N == 0, "mused_swfd_begin_epoch: rows_seen must be a multiple of N");
  hipStream_t st = (hipStream_t)stream;
  const size_t S = h->S, L = h->L;
  MUSED_CHECK_HIP(hipMemsetAsync(h->buf, 0, 8 * S * (size_t)h->n2 * h->d, st));
  MUSED_CHECK_HIP(hipMemsetAsync(h->meta, 0, 4 * S * 4, st));
  MUSED_CHECK_HIP(hipMemsetAsync(h->dropped, 0, 8 * S, st));
  MUSED_CHECK_HIP(hipMemsetAsync(h->qt, 0, 8 * S * h->cap, st));
  (void)L;
  if (main_half) {
    MUSED_REQUIRE(h->lanes == 1, "mused_swfd_begin_epoch: a state blob can only be imported into a single-lane handle");
    int rc = half_copy(h, 0, (char*)main_half, false, st);
    if (rc) return rc;
  }
  h->i = rows_seen;
  h->pend = 0;
  h->restart_mark = rows_seen;  // the epoch-start swap is what this call just did
  h->twin = (main_half == nullptr);
  return MUSED_OK;
}

}  // extern "C"

// Unit-testable primitive: ONE Frequent-Directions rotation of a (2l x d) fp64 buffer in place
// (the FD primitive of SURVEY Appendix A): rows <- sqrt(max(s^2 - s[l-1]^2, 0)) * Vt for the top l
// directions (rows whose shrunk energy is <= 1e-10 s[0]^2 dropped, survivors packed first), rest
// zeroed.  sigma_out (l doubles) receives the norms of the first l rows afterwards.  Builds and
// frees its workspace per call: test/diagnostic use.
extern "C" int mused_fd_rotate(double* buf, int ell, int d, double* sigma_out, int sweeps, void* stream) {
  MUSED_REQUIRE(buf && sigma_out && ell >= 1 && d >= 1, "mused_fd_rotate: bad arguments");
  void* hv = nullptr;
  int rc = mused_swfd_create(1l << 30, 1.0, d, ell, sweeps, &hv);  // one level, theta = 2^30 / l: never dumps
  if (rc) return rc;
  mused::Swfd* h = (mused::Swfd*)hv;
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = sizeof(double) * (size_t)2 * ell * d;
  hipError_t e = hipMemcpyAsync(h->buf, buf, bytes, hipMemcpyDeviceToDevice, st);
  h->i = 2 * ell;
  if (e == hipSuccess) rc = mused::swfd_rotate_all(h, st);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(buf, h->buf, bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && !rc) {
    hipLaunchKernelGGL(mused::swfd_finish_rows_kernel, dim3(ell), dim3(256), 0, st, buf, d, sigma_out);
    e = hipStreamSynchronize(st);
  }
  mused_swfd_destroy(hv);
  if (rc) return rc;
  MUSED_CHECK_HIP(e);
  return MUSED_OK;
}
