// Batched symmetric eigensolver for the small dense eigenproblems of the path: the 2l x 2l Gram matrix of an FD
// rotation (the "per-window SVD/rotation step" of SWFD), the <= 4l x 4l Gram of a sketch query and the
// (l + 10) x (l + 10) Gram of the randomized-SVD projection B = Q^T A.  One-sided (Hestenes) block Jacobi in fp64,
// register resident, orders up to 1024 -- as a hipGraph of block-pair rounds or as one persistent work-queue launch.
#include <stdlib.h>

#include <vector>

#include "internal.h"
#include "wave_ops.h"

namespace mused {

std::recursive_mutex& capture_mutex() {
  static std::recursive_mutex m;
  return m;
}

struct OsjqCtl {
  unsigned head, tail;  // tickets handed to consumers / slots handed to producers
  int finished, nmat;   // matrices whose solve has ended / matrices that take part (representatives)
  int all_done, error;  // set once: nothing will ever be queued again / a consumer gave up waiting (timeout)
  int dirty;            // the unit ring holds entries of the last solve (it is only cleared when it does)
  int pad;
};

struct EigPlan {
  int n, batch, sweeps;
  int method;  // 0 = two-sided, one launch per rotation set; 1 = one-sided block Jacobi (default)
  int ldn;     // OSJ: padded order (multiple of 64) = threads per workgroup
  double* Gc;  // OSJ: batch x ldn x ldn, column-major working copy
  double* lam; // OSJ: batch x ldn column norms
  int* notconv;  // OSJ: (sweeps * rps) x batch flags: a pair that still matters was met in that launch round
  int rps;       // launch rounds per sweep: nb - 1 (orders <= 256: round 0 carries the pairs inside the blocks) or nb
  int* qclean;   // queue solver: consecutive clean rounds per matrix
  int wavek;     // OSJ: 1 = wave-private kernel (orders <= 256)
  bool direct;   // OSJ: the caller fills Gc itself (n == ldn): no pack pass
  const int* rep; // optional: matrix b is a duplicate of matrix rep[b] != b and is not solved
  int sortcols;  // OSJ: store columns by descending norm inside each block pair
  int sort_from; // OSJ wave kernel: first sweep that sorts
  // persistent work-queue solver (orders <= 256, see osjq_kernel): one launch per solve instead of sweeps x (nb - 1)
  int use_queue;
  unsigned* q;        // unit ring: 0 = empty, else 1 + ((matrix * 256 + global round) * 4 + block pair)
  unsigned qcap;
  struct OsjqCtl* qctl;
  // direct solver for the leading pairs only: runs first, the Jacobi then takes the matrices it rejected.
  // 1: orders <= 256 (trd.hip; fallback = the persistent queue Jacobi); 2: orders 320 .. 1024 (trdx.hip; fallback = the sweep graph)
  int trd;
  const int* jrep;     // what the Jacobi's launch-per-round kernels test (matrix b runs when jrep[b] == b): rep, or -- behind the
                       // direct solver of orders 320 .. 1024 (trdx.hip) -- the per-solve list of the matrices it rejected
  int *trdx_act, *trdx_jrep;  // trdx: matrices that passed (act[b] == b) / that the Jacobi must solve (jrep[b] == b)
  int* trdx_nrej;             // trdx: device count of the matrices the last solve rejected
  int* trdx_nrej_host;        // ... its pinned host copy: the Jacobi's sweep graph (hundreds of launches that would find nothing
  hipEvent_t trdx_ev;         //     to do) is only launched when the count is not zero
  int trd_need;        // leading eigenpairs the caller reads
  int trd_cert_all;    // every one of them must pass the certificate (eigenstep) / those that survive the FD shrink
  double* trd_ws;
  int* trd_done;       // per matrix: 1 = solved by the direct solver (the Jacobi skips it)
  unsigned long long q_timeout;  // ticks of s_memrealtime (100 MHz) a consumer waits for its ticket (3 s; MUSED_EIG_QUEUE_TIMEOUT_TICKS)
  int* err_out;                  // optional device word the caller reads back: set when the queue solver gave up (results invalid)
  int* qdone;         // per matrix: units of its current round that have finished
  double* trace; // OSJ adaptive: trace(G) per matrix (owns the allocation notconv / work point into)
  unsigned long long* work;  // OSJ adaptive, profiling: (matrix, sweep) pairs that did work
  // live profiling (off by default): HIP events around every replay of the sweep graph
  bool prof;
  int prof_n;
  std::vector<hipEvent_t>* ev0;
  std::vector<hipEvent_t>* ev1;
  std::vector<hipEvent_t>* evm;  // direct solver: after its first kernel (the tridiagonalisation)
  double* G[2];
  double* V[2];
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStream_t cap_stream;
  bool have_graph;
};

// ======================= one-sided block Jacobi (register resident) =========================
// For a symmetric PSD G = U diag(lam) U^T, orthogonalising the COLUMNS of G by plane rotations
// (Hestenes) ends with columns w_j = lam_j u_j: eigenvalues are the column norms, eigenvectors the
// normalised columns -- no eigenvector accumulation.  Columns are grouped in blocks of CB; one
// workgroup owns a PAIR of blocks for one round (round-robin over block pairs: nb - 1 launches per
// sweep, nb / 2 independent workgroups per matrix and launch).  Thread r keeps ROW r of the 2 CB
// columns in registers (2 CB doubles), runs one full round-robin sweep over those columns with a
// compile-time pairing schedule (register indices are constants), and needs only the CB dot
// products of each step reduced across the workgroup (in-wave transpose-reduce + one LDS hop).
// Column norms are carried in LDS and updated from the rotation, so each step reduces CB values.
constexpr int OSJ_CB = 32;
// Adaptive sweeps: the solve of a matrix ends once a sweep's worth of consecutive launch rounds (every column pair
// once; osj_finished) started no rotation from a column pair that still matters.  With tr = trace(G) (>= lam_max), columns g_j = lam_j u_j, pq = g_p . g_q, a pair matters when
//   cos^2 = pq^2 / (pp qq) > cos2         (what a sweep of smaller cosines leaves behind is second order), and
//   min(pp, qq) > (1e-12 tr)^2            (columns in the numerical null space never settle relatively).
// cos2 = 1e-10 in the wave-private kernel: with its block-pair sort, eigenvectors come out orthogonal to 2e-10
// even inside exactly multiple eigenvalues (tests: test_syevj_special_matrices).  The row-per-thread kernel
// leaves ~1e-2 of the last cosine inside such a cluster and uses 1e-14 (1e-9).
// Tried and dropped: an absolute test (|pq| against (lam_p + lam_q) * mean eigenvalue) and a tighter threshold
// for pairs with |qq - pp| < 4 |pq| -- neither changed a decision on the matrices of this path.
constexpr double OSJ_CONV_COS2_WAVE = 1e-10, OSJ_CONV_COS2_ROW = 1e-14;
__device__ __forceinline__ bool osj_pair_active(double pq2, double pp, double qq, double floor2, double cos2) {
  return (pq2 > cos2 * (pp * qq)) & (pp > floor2) & (qq > floor2);
}

// ROLLING STOP (round 2): flags are kept per LAUNCH ROUND (fr = sweep * rps + position in the sweep; rps = launches per
// sweep), and a matrix stops as soon as the last rps rounds -- any rps consecutive rounds of the cyclic schedule meet
// every column pair exactly once -- raised no flag, instead of waiting for a clean sweep that starts at a sweep boundary
// (on average half a sweep less).
__device__ __forceinline__ bool osj_finished(const int* __restrict__ notconv, int fr, int rps, int batch, int m) {
  if (!notconv || fr < rps) return false;
  int any = 0;
  for (int i = 1; i <= rps; ++i) any |= notconv[(long)(fr - i) * batch + m];
  return any == 0;
}

__host__ __device__ constexpr int osj_pair_p(int m2, int step, int k) {
  // round robin on m2 (even) players; returns the smaller index of pair k at `step`
  const int m = m2 - 1;
  const int a = (k == 0) ? m : (step + k) % m;
  const int b = (k == 0) ? (step % m) : ((step - k + m) % m);
  return a < b ? a : b;
}
__host__ __device__ constexpr int osj_pair_q(int m2, int step, int k) {
  const int m = m2 - 1;
  const int a = (k == 0) ? m : (step + k) % m;
  const int b = (k == 0) ? (step % m) : ((step - k + m) % m);
  return a < b ? b : a;
}

// Sum N values (N = 8, 16, 32) over the 64 lanes of a wave with N - 1 + (6 - log2 N) shuffles: each
// halving stage exchanges half of the values with the lane `mask` away.  On return v[0] of lane l is
// the wave total of value `idx` (lanes that share idx hold the same total).
// ---- wave-level "transpose reduce": N values per lane summed over the 64 lanes --------------------
// Stage partners (all VALU, no LDS crossbar): l^32 by v_permlane32_swap, l^16 by v_permlane16_swap
// (both exchange a kept and a sent value in one instruction, no selects), then row_mirror (flips bit
// 3), row_half_mirror (flips bit 2), quad_perm xor 2, xor 1 through DPP moves.  While more than one
// value is left a stage halves the count: lanes with the stage bit set keep the upper half of the
// values, the others the lower half.  On return lane l holds the wave total of value
// idx = (bits 5, 4, 3, 2[, 1] of l, as many as there were halving stages); lanes sharing idx agree.
template <int N>
__device__ __forceinline__ double wave_treduce(double (&v)[N], int lane, int& idx) {
  static_assert(N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "wave_treduce: N must be 4, 8, 16, 32 or 64");
  if constexpr (N == 4) {
    // two halving stages (l ^ 32, l ^ 16), then the 16 lanes of a row are summed: idx = (bit 5, bit 4)
    v[0] = swap32_add(v[0], v[2]);
    v[1] = swap32_add(v[1], v[3]);
    double t4 = swap16_add(v[0], v[1]);
    t4 = t4 + dpp_mov_f64<DPP_ROW_MIRROR>(t4);
    t4 = t4 + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(t4);
    t4 = t4 + dpp_mov_f64<DPP_QUAD_XOR2>(t4);
    t4 = t4 + dpp_mov_f64<DPP_QUAD_XOR1>(t4);
    idx = ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
    return t4;
  } else {
  // stage 1: l ^ 32 (N -> N/2)
#pragma unroll
  for (int i = 0; i < N / 2; ++i) v[i] = swap32_add(v[i], v[i + N / 2]);
  // stage 2: l ^ 16 (N/2 -> N/4)
#pragma unroll
  for (int i = 0; i < N / 4; ++i) v[i] = swap16_add(v[i], v[i + N / 4]);
  // stage 3: row mirror, decided by bit 3 (N/4 -> N/8)
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      const double keep = hi ? v[i + N / 8] : v[i];
      const double send = hi ? v[i] : v[i + N / 8];
      v[i] = keep + dpp_mov_f64<DPP_ROW_MIRROR>(send);
    }
  }
  int id = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
  double t;
  if constexpr (N == 8) {
    t = v[0] + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v[0]);
    t = t + dpp_mov_f64<DPP_QUAD_XOR2>(t);
    t = t + dpp_mov_f64<DPP_QUAD_XOR1>(t);
  } else {
    {  // stage 4: half mirror, decided by bit 2 (N/8 -> N/16)
      const bool hi = (lane & 4) != 0;
#pragma unroll
      for (int i = 0; i < N / 16; ++i) {
        const double keep = hi ? v[i + N / 16] : v[i];
        const double send = hi ? v[i] : v[i + N / 16];
        v[i] = keep + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(send);
      }
      id = id * 2 + ((lane >> 2) & 1);
    }
    if constexpr (N == 16) {
      t = v[0] + dpp_mov_f64<DPP_QUAD_XOR2>(v[0]);
      t = t + dpp_mov_f64<DPP_QUAD_XOR1>(t);
    } else {
      {  // stage 5: xor 2, decided by bit 1 (N/16 -> N/32)
        const bool hi = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < N / 32; ++i) {
          const double keep = hi ? v[i + N / 32] : v[i];
          const double send = hi ? v[i] : v[i + N / 32];
          v[i] = keep + dpp_mov_f64<DPP_QUAD_XOR2>(send);
        }
        id = id * 2 + ((lane >> 1) & 1);
      }
      if constexpr (N == 32) {
        t = v[0] + dpp_mov_f64<DPP_QUAD_XOR1>(v[0]);
      } else {  // N == 64, stage 6: xor 1, decided by bit 0
        const bool hi = (lane & 1) != 0;
        const double keep = hi ? v[1] : v[0];
        const double send = hi ? v[0] : v[1];
        t = keep + dpp_mov_f64<DPP_QUAD_XOR1>(send);
        id = id * 2 + (lane & 1);
      }
    }
  }
  idx = id;
  return t;
  }
}

__device__ __forceinline__ double osj_readlane(double v, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), l);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Plane rotation that zeroes the inner product pq of two columns with squared norms pp, qq:
//   zeta = (qq - pp) / (2 pq),  t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),  c = 1/sqrt(1 + t^2),  s = t c.
// This sits on the critical path of every Jacobi step (one wave-instruction stream, nothing to overlap),
// so the two divides and two square roots are v_rcp_f64 / v_rsq_f64 seeds refined by Newton steps to
// full fp64 accuracy (an angle error d would leave d * sqrt(pp / qq) of cosine between a large and a
// small column and stall convergence on graded spectra) instead of the ~4x longer IEEE sequences.
__device__ __forceinline__ double osj_rcp(double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = fma(r, fma(-y, r, 1.0), r);
  r = fma(r, fma(-y, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double osj_rsqrt(double h) {
  double r = __builtin_amdgcn_rsq(h);
  r = r * fma(-0.5 * h, r * r, 1.5);
  r = r * fma(-0.5 * h, r * r, 1.5);
  return r;
}
__device__ __forceinline__ void osj_rotation(double pp, double qq, double pq, double& c, double& s, double& npp,
                                             double& nqq) {
  c = 1.0; s = 0.0; npp = pp; nqq = qq;
  if (pq * pq > 1e-30 * (pp * qq) && pq != 0.0) {
    const double zeta = (qq - pp) * osj_rcp(2.0 * pq);
    const double az = fabs(zeta);
    double t;
    if (!(az < 1e100)) {
      t = 0.0;  // |zeta| astronomically large (or the reciprocal overflowed): the angle is below rounding
    } else {
      const double w = fma(az, az, 1.0);
      const double y = fma(w, osj_rsqrt(w), az);  // |zeta| + sqrt(1 + zeta^2) >= 1
      const double r = osj_rcp(y);
      t = zeta < 0.0 ? -r : r;
    }
    const double h = fma(t, t, 1.0);  // in [1, 2]
    c = osj_rsqrt(h);
    s = t * c;
    npp = pp - t * pq;
    nqq = qq + t * pq;
  }
}

// The same rotation without a branch and with a shorter dependency chain, returning (c, t = s / c) -- what
// the scaled ("fast") rotations need.  With a = qq - pp, b = 2 pq:
//     t = sign(a b) |b| / (|a| + sqrt(a^2 + b^2))          (the small root of t^2 + 2 zeta t - 1 = 0)
// costs one rsq and one rcp; their hardware seeds are good to ~5e-8, ONE Newton step (2e-15) is enough for
// the angle (an angle error only leaves that fraction of the inner product behind), TWO are used for
// c = 1 / sqrt(1 + t^2), which rescales the columns and has to be exact to rounding.
__device__ __forceinline__ void osj_rotation_t(double pp, double qq, double pq, double& c, double& t, double& npp,
                                               double& nqq) {
  const double a = qq - pp, b = pq + pq;
  const double s2 = fma(a, a, b * b);
  const bool rot = (pq * pq > 1e-30 * (pp * qq)) && (s2 > 0.0) && (s2 < 1e300);
  double rs = __builtin_amdgcn_rsq(s2);
  rs = rs * fma(-0.5 * s2, rs * rs, 1.5);
  const double den = fma(s2, rs, fabs(a));
  double rc = __builtin_amdgcn_rcp(den);
  rc = fma(rc, fma(-den, rc, 1.0), rc);
  double tt = fabs(b) * rc;
  tt = ((a < 0.0) != (b < 0.0)) ? -tt : tt;
  t = rot ? tt : 0.0;
  const double h = fma(t, t, 1.0);  // in [1, 2]
  const double cc = osj_rsqrt(h);
  c = rot ? cc : 1.0;
  npp = fma(-t, pq, pp);
  nqq = fma(t, pq, qq);
}

// Pairing schedules of the columns held by one workgroup (compile-time: register indices are constants).
//   CROSS = false: round robin over all C2 columns (C2 - 1 steps of C2 / 2 pairs): every pair once.
//   CROSS = true : only pairs (i, CB + (i + step) % CB) between the two blocks (CB steps of CB pairs);
//                  pairs inside a block are rotated once per sweep by the INTRA launch instead, so a
//                  sweep touches every pair of the matrix exactly once (n - 1 dependent steps).
template <bool CROSS>
__host__ __device__ constexpr int osj_sched_p(int c2, int step, int k) {
  return CROSS ? k : osj_pair_p(c2, step, k);
}
template <bool CROSS>
__host__ __device__ constexpr int osj_sched_q(int c2, int step, int k) {
  return CROSS ? (c2 / 2 + (k + step) % (c2 / 2)) : osj_pair_q(c2, step, k);
}

template <int CB, int NT, int MODE, int DBG = 0>
__global__ __launch_bounds__(NT) void osj_round_kernel(double* __restrict__ Gc, int n, int ldn, int nb, int round,
                                                      int* __restrict__ notconv, int fr, int rps,
                                                      const double* __restrict__ trace, int sortcols,
                                                      const int* __restrict__ rep) {
  if (rep && rep[blockIdx.y] != (int)blockIdx.y) return;  // duplicate of another matrix of the batch
  // ADAPTIVE SWEEPS: notconv[sweep * batch + matrix] is set by any workgroup that met, in this sweep, a
  // column pair that still matters (osj_pair_active) (columns converge
  // to lambda_j u_j: pairs inside the numerical null space never settle in the relative sense and carry
  // nothing the path uses); a matrix whose previous sweep set nothing is converged (that sweep left
  // every such cosine below ~1e-13 * n by quadratic convergence) and its workgroups return at once.
  if (osj_finished(notconv, fr, rps, gridDim.y, blockIdx.y)) return;
  double small2 = 0.0;
  if (trace) {
    const double tr = trace[blockIdx.y];
    small2 = 1e-24 * tr * tr;
  }
  // MODE 1: block pair (bp, bq) of the round-robin, cross pairs only; MODE 2: ONE block of 2*CB
  // consecutive columns (blockIdx.x), all pairs inside it; MODE 0: block pair, all pairs.
  // DBG != 0: timing-only ablations (results are wrong): 2 no rotation maths, 3 no barrier, 4 no apply.
  //
  // FAST (scaled) ROTATIONS: column j is held as d_j * x_j with a per-column scale d_j (LDS, starts at
  // 1).  A rotation (c, s, t = s/c) of the true columns becomes
  //     x_p <- x_p - (t d_q / d_p) x_q,   x_q <- x_q + (t d_p / d_q) x_p,   d_p, d_q <- c d_p, c d_q
  // i.e. TWO fmas per element instead of four multiply-adds; 1/d_j is carried too (1/c = c (1 + t^2)),
  // so no divide.  Scales are folded back into the columns when the launch stores them (<= 2*CB - 1
  // rotations per column per launch: d >= 2^-16, no underflow).
  constexpr bool CROSS = (MODE == 1);
  constexpr int C2 = 2 * CB;
  constexpr int NSTEP = (DBG == 5 || DBG == 6) ? 0 : (DBG == 7 ? 4 : (CROSS ? CB : C2 - 1));
  constexpr int NW = NT / 64;
  __shared__ double part[2][NW][C2];  // double-buffered cross-wave partial sums
  __shared__ double nrm[NW][C2];      // per-wave private copies: true squared column norms,
  __shared__ double dsc[NW][C2];      //   column scales d_j
  __shared__ double isc[NW][C2];      //   and 1 / d_j
  __shared__ double2 tau[NW][CB];     // per-wave (tau_p, tau_q) of the step's CB pairs
  __shared__ int dest[NW][C2];        // store position of each column (descending norm)

  double* M = Gc + (long)blockIdx.y * ldn * ldn;
  int bp, bq;
  if (MODE == 2) {
    bp = 2 * blockIdx.x;
    bq = bp + 1;
  } else {
    const int k = blockIdx.x, m = nb - 1;
    int a = (k == 0) ? m : (round + k) % m;
    int b = (k == 0) ? (round % m) : ((round - k + m) % m);
    bp = a < b ? a : b;
    bq = a < b ? b : a;
  }
  const int r = threadIdx.x, lane = r & 63, wave = r >> 6;
  double x[C2];
  int active = 0;
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    x[j] = M[(long)(bp * CB + j) * ldn + r];
    x[CB + j] = M[(long)(bq * CB + j) * ldn + r];
  }
  // squared column norms (every wave ends up with its own full copy)
  {
    double sq[C2];
#pragma unroll
    for (int j = 0; j < C2; ++j) sq[j] = x[j] * x[j];
    int idx;
    const double t = wave_treduce<C2>(sq, lane, idx);
    if ((lane & ((64 / C2) - 1)) == 0) part[1][wave][idx] = t;
    __syncthreads();
    if (lane < C2) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += part[1][w][lane];
      nrm[wave][lane] = sum;
      dsc[wave][lane] = 1.0;
      isc[wave][lane] = 1.0;
    }
    // part[1] is next written at step 1, after the barrier of step 0: no hazard
  }
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int buf = step & 1;
    double dv[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) dv[k] = x[osj_sched_p<CROSS>(C2, step, k)] * x[osj_sched_q<CROSS>(C2, step, k)];
    int idx;
    const double t = wave_treduce<CB>(dv, lane, idx);
    if ((lane & ((64 / CB) - 1)) == 0) part[buf][wave][idx] = t;
    if constexpr (DBG != 3) __syncthreads();
    if (lane < CB) {  // every wave computes all CB rotations redundantly: no second barrier
      double raw = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) raw += part[buf][w][lane];
      int p, q;
      if (CROSS) {
        p = lane;
        q = CB + (lane + step) % CB;
      } else {
        const int m = C2 - 1;
        const int a = (lane == 0) ? m : (step + lane) % m;
        const int b = (lane == 0) ? (step % m) : ((step - lane + m) % m);
        p = a < b ? a : b;
        q = a < b ? b : a;
      }
      const double dp = dsc[wave][p], dq = dsc[wave][q];
      const double pq = raw * dp * dq;
      active |= osj_pair_active(pq * pq, nrm[wave][p], nrm[wave][q], small2, OSJ_CONV_COS2_ROW);
      double c = 1.0, tt = 0.0, npp, nqq;
      if constexpr (DBG == 2) {
        c = 0.8; tt = 0.75; npp = nrm[wave][p] + pq; nqq = nrm[wave][q];
      } else {
        osj_rotation_t(nrm[wave][p], nrm[wave][q], pq, c, tt, npp, nqq);
      }
      double tp = 0.0, tq = 0.0;
      if (tt != 0.0) {
        const double ip = isc[wave][p], iq = isc[wave][q];
        tp = tt * dq * ip;
        tq = tt * dp * iq;
        const double ic = c * (1.0 + tt * tt);
        nrm[wave][p] = npp;
        nrm[wave][q] = nqq;
        dsc[wave][p] = dp * c;
        dsc[wave][q] = dq * c;
        isc[wave][p] = ip * ic;
        isc[wave][q] = iq * ic;
      }
      tau[wave][lane] = make_double2(tp, tq);
    }
    // same wave wrote tau: program order + the compiler's lgkmcnt wait make it visible (no barrier)
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const double2 t2 = tau[wave][k];
      const int p = osj_sched_p<CROSS>(C2, step, k), q = osj_sched_q<CROSS>(C2, step, k);
      const double xp = x[p], xq = x[q];
      if constexpr (DBG == 4) {
        x[p] = xp + t2.x;
      } else {
        x[p] = fma(-t2.x, xq, xp);
        x[q] = fma(t2.y, xp, xq);
      }
    }
  }
  // de Rijk at block level: store the columns by descending norm (large columns first speeds up the
  // convergence on graded spectra); fold the scales back in.
  if (lane < C2) {
    const double mine = nrm[wave][lane];
    int rank = 0;
#pragma unroll
    for (int j = 0; j < C2; ++j) {
      const double o = nrm[wave][j];
      rank += (o > mine) || (o == mine && j < lane);
    }
    dest[wave][lane] = sortcols ? rank : lane;
  }
#pragma unroll
  for (int j = 0; j < C2; ++j) {
    const int pos = dest[wave][j];
    const int col = (pos < CB) ? (bp * CB + pos) : (bq * CB + pos - CB);
    M[(long)col * ldn + r] = x[j] * dsc[wave][j];
  }
  if (notconv && wave == 0 && __any(active) && lane == 0) atomicOr(&notconv[(long)fr * gridDim.y + blockIdx.y], 1);
  (void)n;
}

// ---- wave-private variant (orders <= 256) -----------------------------------------------------------
// Same sweep (every column pair once: pairs inside a block first, then the block-pair rounds), but the 64
// columns of a workgroup are dealt to its 4 waves as 8 SETS of 8 columns, two sets per wave, and lane l
// of every wave holds rows l, l + 64, ... (RP = n / 64 rows) of its 16 columns.  All 64 pairs between a
// wave's two sets (8 steps of 8 disjoint pairs) and the pairs inside them (7 steps) are then wave-private
// work: RP products per pair are summed in the lane, ONE 8-value transpose-reduce finishes the dot
// products inside the wave (the row-per-thread kernel above reduces 32 values per step and needs an LDS hop
// plus a workgroup barrier for the cross-wave sum), every lane computes the rotation of "its" pair, and
// the waves run without any barrier until the sets are dealt again: one set per wave goes through LDS
// (scales folded in) at each of the 3 (block-pair round) or 6 (round 0: pairs inside the two blocks
// first) re-deals of a launch.
//   sets 0-3 = columns of block bp, 4-7 = block bq.  Re-deal table (A = fixed slot, B = exchanged slot;
//   "swap" exchanges a wave's slots first):
//     round 0 :  (0,1)(2,3)(4,5)(6,7)  intra-set pairs, then A x B
//                swap w1,w3; B: w0<->w1, w2<->w3   -> (0,2)(3,1)(4,6)(7,5)
//                swap w1,w3; B: w0<->w1, w2<->w3   -> (0,3)(1,2)(4,7)(5,6)     all pairs inside bp, bq done
//                B: w0<->w2, w1<->w3               -> (0,7)(1,6)(4,3)(5,2)
//                swap w2,w3; B ring w <- w+1       -> (0,6)(1,4)(3,5)(2,7) -> (0,4)(1,5)(3,7)(2,6) -> (0,5)(1,7)(3,6)(2,4)
//     round r :  (0,4)(1,5)(2,6)(3,7), then three times B ring w <- w+1.
// SC = columns per set, NW = OSJ_CB / SC = waves per workgroup.  Orders <= 256: SC = 8, NW = 4 (all of the above).
// Orders 320-512 hold 5-8 rows per lane, so a wave can keep only 2 x 4 columns: SC = 4, NW = 8, 4 pairs per step,
// 8 phases of 4 steps per block-pair round; the pairs inside the blocks are left to the row-per-thread kernel's
// intra-block launch (PREFIX is only built for SC = 8).

template <int RP, int SC>
struct OsjwShared {
  static constexpr int NW = OSJ_CB / SC;
  double xfer[NW][SC * RP][64];  // one set per wave in flight
  double xnrm[NW][SC];
  int xid[NW];
  double nrm[NW][2 * SC];  // true squared norms of the wave's 16 columns
  double dsc[NW][2 * SC];  // column scales d_j (column held as d_j * x_j)
  double isc[NW][2 * SC];  // 1 / d_j
  double2 tau[NW][SC];
};

// one step: 8 disjoint pairs (P(k), Q(k)) of the wave's 16 columns (compile-time indices)
template <int RP, int SC, bool INTRA, int T>
__device__ __forceinline__ void osjw_step(double (&x)[2 * SC * RP], OsjwShared<RP, SC>& sh, int wave, int lane,
                                          double small2, int& active) {
  constexpr int H = SC / 2;  // pairs inside one set per intra step
  auto P = [](int k) constexpr { return INTRA ? (k / H) * SC + osj_pair_p(SC, T, k % H) : k; };
  auto Q = [](int k) constexpr { return INTRA ? (k / H) * SC + osj_pair_q(SC, T, k % H) : SC + (k + T) % SC; };
  // this lane's pair (known before the dot products: the reads of its scalars overlap them)
  const int idx = (SC == 8) ? ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)
                            : ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);  // as wave_treduce<SC> deals them
  int lp, lq;
  int tv = T;  // opaque copy: the (lane-dependent) LDS addresses are recomputed per step (3 VALU ops) instead
  asm volatile("" : "+s"(tv));  //   of being kept alive -- and spilled -- across the phases of a launch
  if (INTRA) {
    const int m = SC - 1, kk = idx % H;
    const int a = (kk == 0) ? m : (tv + kk) % m;
    const int b = (kk == 0) ? (tv % m) : ((tv - kk + m) % m);
    lp = (idx / H) * SC + (a < b ? a : b);
    lq = (idx / H) * SC + (a < b ? b : a);
  } else {
    lp = idx;
    lq = SC + ((idx + tv) & (SC - 1));
  }
  const double dp = sh.dsc[wave][lp], dq = sh.dsc[wave][lq];
  const double pp = sh.nrm[wave][lp], qq = sh.nrm[wave][lq];
  const double ip = sh.isc[wave][lp], iq = sh.isc[wave][lq];
  double dv[SC];
#pragma unroll
  for (int k = 0; k < SC; ++k) {
    const int p = P(k), q = Q(k);
    double acc = x[p * RP] * x[q * RP];
#pragma unroll
    for (int i = 1; i < RP; ++i) acc = fma(x[p * RP + i], x[q * RP + i], acc);
    dv[k] = acc;
  }
  int idx2;
  const double raw = wave_treduce<SC>(dv, lane, idx2);  // every lane: the dot product of pair idx
  {
    const double pq = raw * dp * dq;
    // small2 < 0: fixed sweep count, no convergence flag (kept as a branch per step: left unconditional the
    // compiler sinks all comparisons of a launch to its end and keeps their operands alive until then)
    if (small2 >= 0.0) active |= osj_pair_active(pq * pq, pp, qq, small2, OSJ_CONV_COS2_WAVE);
    double c, tt, npp, nqq;
    osj_rotation_t(pp, qq, pq, c, tt, npp, nqq);
    const double ic = c * fma(tt, tt, 1.0);  // 1 / c
    if ((lane & (64 / SC - 1)) == 0) {
      sh.nrm[wave][lp] = npp;
      sh.nrm[wave][lq] = nqq;
      sh.dsc[wave][lp] = dp * c;
      sh.dsc[wave][lq] = dq * c;
      sh.isc[wave][lp] = ip * ic;
      sh.isc[wave][lq] = iq * ic;
      sh.tau[wave][idx] = make_double2(tt * dq * ip, tt * dp * iq);
    }
  }
  // same wave wrote tau: LDS operations of a wave complete in order
#pragma unroll
  for (int k = 0; k < SC; ++k) {
    if (RP >= 4 && k == SC / 2) __builtin_amdgcn_sched_barrier(0);  // two batches of tau reads: fewer live registers
    const double2 t2 = sh.tau[wave][k];
    const int p = P(k), q = Q(k);
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const double xp = x[p * RP + i], xq = x[q * RP + i];
      x[p * RP + i] = fma(-t2.x, xq, xp);
      x[q * RP + i] = fma(t2.y, xp, xq);
    }
  }
}

template <int RP, int SC, bool INTRA, int T = 0>
__device__ __forceinline__ void osjw_phase(double (&x)[2 * SC * RP], OsjwShared<RP, SC>& sh, int wave, int lane,
                                           double small2, int& active) {
  constexpr int NSTEP = INTRA ? SC - 1 : SC;
  if constexpr (T < NSTEP) {
    osjw_step<RP, SC, INTRA, T>(x, sh, wave, lane, small2, active);
    osjw_phase<RP, SC, INTRA, T + 1>(x, sh, wave, lane, small2, active);
  }
}

// re-deal: (optionally swap the wave's slots,) hand slot B to LDS and take the set wave `src` handed in
template <int RP, int SC>
__device__ __forceinline__ void osjw_deal(double (&x)[2 * SC * RP], OsjwShared<RP, SC>& sh, int wave, int lane,
                                          bool do_swap, int src, int& ida, int& idb) {
  if (do_swap) {  // wave-uniform
#pragma unroll
    for (int e = 0; e < SC * RP; ++e) {
      const double t = x[e];
      x[e] = x[SC * RP + e];
      x[SC * RP + e] = t;
    }
    if (lane < SC) {
      const double n0 = sh.nrm[wave][lane], d0 = sh.dsc[wave][lane], i0 = sh.isc[wave][lane];
      sh.nrm[wave][lane] = sh.nrm[wave][SC + lane];
      sh.dsc[wave][lane] = sh.dsc[wave][SC + lane];
      sh.isc[wave][lane] = sh.isc[wave][SC + lane];
      sh.nrm[wave][SC + lane] = n0;
      sh.dsc[wave][SC + lane] = d0;
      sh.isc[wave][SC + lane] = i0;
    }
    const int t = ida;
    ida = idb;
    idb = t;
  }
#pragma unroll
  for (int j = 0; j < SC; ++j) {
    const double d = sh.dsc[wave][SC + j];
#pragma unroll
    for (int i = 0; i < RP; ++i) sh.xfer[wave][j * RP + i][lane] = x[(SC + j) * RP + i] * d;
  }
  if (lane < SC) sh.xnrm[wave][lane] = sh.nrm[wave][SC + lane];
  if (lane == 0) sh.xid[wave] = idb;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < SC * RP; ++e) x[SC * RP + e] = sh.xfer[src][e][lane];
  if (lane < SC) {
    sh.nrm[wave][SC + lane] = sh.xnrm[src][lane];
    sh.dsc[wave][SC + lane] = 1.0;
    sh.isc[wave][SC + lane] = 1.0;
  }
  idb = sh.xid[src];
  __syncthreads();  // everyone has taken its set: the buffers may be written again
}

// Loads / stores of the matrix columns.  COH = false: plain accesses (the kernel boundary publishes them).  COH = true:
// agent-scope relaxed atomics = global_load / global_store ... sc1 -- every byte another workgroup of the SAME launch
// wrote is read past this CU's L1 and every byte it will read is written through (persistent solver below).
template <bool COH>
__device__ __forceinline__ double osj_ld(const double* p) {
  if constexpr (COH) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else {
    return *p;
  }
}
template <bool COH>
__device__ __forceinline__ void osj_st(double* p, double v) {
  if constexpr (COH) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// One block-pair round of one matrix by one workgroup (the body of osjw_kernel): returns (per wave) whether a pair that
// still matters was met.  kidx = index of the block pair in the round-robin of `round`.
template <int RP, int SC, bool PREFIX, bool COH, bool ST_WT = COH>
__device__ __forceinline__ int osjw_unit(double* __restrict__ M, int ldn, int nb, int round, int kidx, double small2,
                                         int sortcols, OsjwShared<RP, SC>& sh, int tid) {
  static_assert(!PREFIX || SC == 8, "the in-launch intra-block schedule exists for 8-column sets only");
  constexpr int CB = OSJ_CB, NW = OSJ_CB / SC;
  int bp, bq;
  {
    const int k = kidx, m = nb - 1;
    const int a = (k == 0) ? m : (round + k) % m;
    const int b = (k == 0) ? (round % m) : ((round - k + m) % m);
    bp = a < b ? a : b;
    bq = a < b ? b : a;
  }
  const int lane = tid & 63, wave = tid >> 6;
  // sets 0 .. NW-1 = columns of block bp, NW .. 2NW-1 = block bq
  int ida = PREFIX ? 2 * wave : wave, idb = PREFIX ? 2 * wave + 1 : NW + wave;
  auto set_col = [&](int sid) { return (sid < NW ? bp * CB + SC * sid : bq * CB + SC * (sid - NW)); };
  double x[2 * SC * RP];
  {
    const int ca = set_col(ida), cb = set_col(idb);
#pragma unroll
    for (int j = 0; j < SC; ++j)
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        x[j * RP + i] = osj_ld<COH>(&M[(long)(ca + j) * ldn + lane + 64 * i]);
        x[(SC + j) * RP + i] = osj_ld<COH>(&M[(long)(cb + j) * ldn + lane + 64 * i]);
      }
  }
  {  // squared norms of the wave's 2 SC columns
    double sq[2 * SC];
#pragma unroll
    for (int c = 0; c < 2 * SC; ++c) {
      double acc = x[c * RP] * x[c * RP];
#pragma unroll
      for (int i = 1; i < RP; ++i) acc = fma(x[c * RP + i], x[c * RP + i], acc);
      sq[c] = acc;
    }
    int idx;
    const double t = wave_treduce<2 * SC>(sq, lane, idx);
    if ((lane & (64 / (2 * SC) - 1)) == 0) {
      sh.nrm[wave][idx] = t;
      sh.dsc[wave][idx] = 1.0;
      sh.isc[wave][idx] = 1.0;
    }
  }
  int active = 0;
  if constexpr (PREFIX) {
    osjw_phase<RP, SC, true>(x, sh, wave, lane, small2, active);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, (wave & 1) != 0, wave ^ 1, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, (wave & 1) != 0, wave ^ 1, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, false, wave ^ 2, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, wave >= 2, (wave + 1) & 3, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, false, (wave + 1) & 3, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
    osjw_deal<RP, SC>(x, sh, wave, lane, false, (wave + 1) & 3, ida, idb);
    osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
  } else {
    // NW phases (A x B of the wave's two sets), the B sets moving round the ring of waves in between: every set of
    // bp meets every set of bq.
#pragma unroll
    for (int ph = 0; ph < NW; ++ph) {
      osjw_phase<RP, SC, false>(x, sh, wave, lane, small2, active);
      if (ph + 1 < NW) osjw_deal<RP, SC>(x, sh, wave, lane, false, (wave + 1) & (NW - 1), ida, idb);
    }
  }
  if (sortcols && lane == 0) sh.xid[wave] = (ida << 8) | idb;  // (xid is free after the last re-deal)
  if (sortcols) {
    // de Rijk at block-pair level: the 64 columns go back in descending norm (ties: current position).  Costs
    // about one sweep on well-separated spectra and is what makes clustered / multiple eigenvalues converge.
    __syncthreads();  // all waves' norms are final
    // lane l stands for column (wave l / 2SC, slot l % 2SC) of the workgroup: one ballot per own column gives its rank
    const int ow = lane / (2 * SC), oc = lane % (2 * SC);
    const double other = sh.nrm[ow][oc];
    const int oid = ((oc < SC) ? (sh.xid[ow] >> 8) : (sh.xid[ow] & 255)) * SC + (oc & (SC - 1));
#pragma unroll
    for (int c = 0; c < 2 * SC; ++c) {
      const double mine = sh.nrm[wave][c];
      const int myid = (c < SC ? ida : idb) * SC + (c & (SC - 1));
      const int pos = __popcll(__ballot((other > mine) || (other == mine && oid < myid)));
      const int colg = (pos < CB) ? (bp * CB + pos) : (bq * CB + pos - CB);
      const double dd = sh.dsc[wave][c];
#pragma unroll
      for (int i = 0; i < RP; ++i) osj_st<ST_WT>(&M[(long)colg * ldn + lane + 64 * i], x[c * RP + i] * dd);
    }
  } else {
    const int ca = set_col(ida), cb = set_col(idb);
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const double da = sh.dsc[wave][j], db = sh.dsc[wave][SC + j];
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        osj_st<ST_WT>(&M[(long)(ca + j) * ldn + lane + 64 * i], x[j * RP + i] * da);
        osj_st<ST_WT>(&M[(long)(cb + j) * ldn + lane + 64 * i], x[(SC + j) * RP + i] * db);
      }
    }
  }
  return active;
}

template <int RP, int SC, bool PREFIX>
__global__ __launch_bounds__(64 * (OSJ_CB / SC), SC == 8 ? 2 : 1) void osjw_kernel(double* __restrict__ Gc, int ldn, int nb,
                                                                                  int round,
                                                                                  int* __restrict__ notconv, int fr, int rps,
                                                                                  const double* __restrict__ trace,
                                                                                  int sortcols,
                                                                                  const int* __restrict__ rep) {
  if (rep && rep[blockIdx.y] != (int)blockIdx.y) return;  // duplicate of another matrix of the batch
  if (osj_finished(notconv, fr, rps, gridDim.y, blockIdx.y)) return;  // see osj_round_kernel
  double small2 = notconv ? 0.0 : -1.0;
  if (trace) {
    const double tr = trace[blockIdx.y];
    small2 = 1e-24 * tr * tr;
  }
  __shared__ OsjwShared<RP, SC> sh;
  double* M = Gc + (long)blockIdx.y * ldn * ldn;
#ifdef MUSED_OSJ_WT_STORES
  const int active = osjw_unit<RP, SC, PREFIX, false, true>(M, ldn, nb, round, blockIdx.x, small2, sortcols, sh, threadIdx.x);
#else
  const int active = osjw_unit<RP, SC, PREFIX, false>(M, ldn, nb, round, blockIdx.x, small2, sortcols, sh, threadIdx.x);
#endif
  if (notconv && __any(active) && (threadIdx.x & 63) == 0) atomicOr(&notconv[(long)fr * gridDim.y + blockIdx.y], 1);
}

// ---- persistent work-queue solver (orders <= 256) ---------------------------------------------------------------
// The launch-per-round solver above pays, per block-pair round, a kernel boundary whose length is set by the slowest
// matrix of the batch, and runs as many rounds as the slowest matrix needs sweeps.  Here ONE launch solves the whole
// batch: a UNIT of work is the same block-pair round of one matrix (osjw_unit), units are handed out through a ticket
// queue in global memory, and a matrix advances on its own -- the workgroup that finishes the last unit of a matrix's
// round r queues the units of round r + 1 (or ends the matrix: converged / sweep cap).  Matrices that converge early
// stop costing anything and the workgroup slots stay busy until the batch is done.
//   * No workgroup ever waits for a particular other workgroup to be resident: a consumer holding ticket t waits for
//     slot t of the queue, which is filled by whichever workgroup completes the round that produces it -- and every
//     such round consists of units already handed to running workgroups.  Any number of resident workgroups >= 1 makes
//     progress, so the kernel cannot deadlock on co-residency (MI355X_MICROARCH: dispatch order is not a contract).
//   * Visibility between workgroups (per-XCD L2s are not coherent, a CU's L1 is never refreshed): every byte of the
//     matrices is written by sc1 write-through stores and read by sc1 loads (osj_ld / osj_st<true>); a unit's waves drain
//     their stores (s_waitcnt vmcnt(0)), meet at the workgroup barrier, then ONE lane counts the unit in with an
//     agent-scope atomic; queue slots, counters and flags are agent-scope atomics throughout.
//   * Every spin is bounded (s_memrealtime): on a timeout the error word is set and every workgroup leaves.
// Arithmetic and order of operations per matrix are those of the launch-per-round solver: results are bit-identical.
constexpr unsigned OSJQ_EXIT = 0xffffffffu;

__global__ void osjq_init_kernel(OsjqCtl* __restrict__ ctl, unsigned* __restrict__ q, int* __restrict__ qdone,
                                 int* __restrict__ qclean, int batch, int upr, const int* __restrict__ rep,
                                 const int* __restrict__ skip) {
  // (q has been zeroed by the launch before)
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m == 0) {
    ctl->head = 0; ctl->finished = 0; ctl->all_done = 0; ctl->error = 0;
  }
  if (m >= batch) return;
  qdone[m] = 0;
  qclean[m] = 0;
  if (rep && rep[m] != m) return;
  if (skip && skip[m]) return;  // already solved by the direct solver
  const unsigned pos = atomicAdd(&ctl->tail, (unsigned)upr);
  for (int i = 0; i < upr; ++i) q[pos + i] = 1u + (((unsigned)m * 256u + 0u) * 4u + (unsigned)i);
  atomicAdd(&ctl->nmat, 1);
  ctl->dirty = 1;  // (benign race: everybody writes 1)
}

__global__ void osjq_reset_kernel(OsjqCtl* __restrict__ ctl) { ctl->tail = 0; ctl->nmat = 0; ctl->dirty = 0; }
template <int RP>
__global__ __launch_bounds__(256, 2) void osjq_kernel(double* __restrict__ Gc, int ldn, int nb, int batch, int max_sweeps,
                                                     int sort_from, int* __restrict__ notconv,
                                                     const double* __restrict__ trace, unsigned* __restrict__ q,
                                                     unsigned qcap, OsjqCtl* __restrict__ ctl, int* __restrict__ qdone,
                                                     int* __restrict__ qclean, unsigned long long timeout,
                                                     int* __restrict__ err_out) {
  __shared__ OsjwShared<RP, 8> sh;
  __shared__ unsigned s_item;
  const int rounds = nb - 1, upr = nb / 2;
  while (true) {
    if (threadIdx.x == 0) {
      unsigned item = OSJQ_EXIT;
      // (nothing queued -- the usual case behind the direct solver, sealed before this launch -- or everything finished:
      // leave without taking a ticket)
      const bool over = __hip_atomic_load(&ctl->all_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      const unsigned t = over ? qcap : atomicAdd(&ctl->head, 1u);
      if (t < qcap) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        while ((item = __hip_atomic_load(&q[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
          if (__hip_atomic_load(&ctl->all_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            item = OSJQ_EXIT;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
          if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {  // (3 s by default) give up, tell everyone
            __hip_atomic_store(&ctl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->all_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (err_out) atomicOr(err_out, 1);  // the caller's status word: the results of this solve are invalid
            item = OSJQ_EXIT;
            break;
          }
        }
      }
      s_item = item;
    }
    __syncthreads();
    const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)s_item);  // wave-uniform: lives in SGPRs
    __syncthreads();  // s_item is rewritten by the next round of the loop
    if (item == OSJQ_EXIT) return;
    const unsigned v = item - 1u;
    const int k = (int)(v & 3u), R = (int)((v >> 2) & 255u), m = (int)(v >> 10);
    const int sweep = R / rounds, r = R - sweep * rounds;
    double small2 = notconv ? 0.0 : -1.0;
    if (trace) {
      const double tr = trace[m];
      small2 = 1e-24 * tr * tr;
    }
    double* M = Gc + (long)m * ldn * ldn;
    const int so = sweep >= sort_from ? 1 : 0;
    int active;
    // the thread index is made opaque per unit: otherwise every lane-dependent LDS / global address of the unit is a
    // loop invariant of the persistent loop, gets hoisted out of it and is kept (and spilled) across the whole unit
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (r == 0) active = osjw_unit<RP, 8, true, true>(M, ldn, nb, 0, k, small2, so, sh, tid);
    else active = osjw_unit<RP, 8, false, true>(M, ldn, nb, r, k, small2, so, sh, tid);
    if (notconv && __any(active) && (threadIdx.x & 63) == 0) atomicOr(&notconv[(long)R * batch + m], 1);
    // publish: every wave drains its write-through stores (and its flag atomic) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ... all of them have; LDS of this unit is free again
    if (threadIdx.x == 0) {  // ... then ONE lane counts the unit in
      const int old = atomicAdd(&qdone[m], 1);
      if (old == upr - 1) {  // last unit of round R of matrix m: nobody else touches this matrix now
        __hip_atomic_store(&qdone[m], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // rolling stop, as osj_finished: the matrix ends with the first round that closes `rounds` clean rounds in a row
        bool fin = (r == rounds - 1) && (sweep + 1 >= max_sweeps);
        if (notconv && !fin) {
          const int flag = __hip_atomic_load(&notconv[(long)R * batch + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (one lane per round touches qclean[m], but rounds of a matrix end on different CUs: agent-scope accesses)
          const int clean = flag ? 0 : __hip_atomic_load(&qclean[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
          __hip_atomic_store(&qclean[m], clean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fin = clean >= rounds;
        }
        if (fin) {
          const int f = atomicAdd(&ctl->finished, 1);
          if (f + 1 == ctl->nmat) __hip_atomic_store(&ctl->all_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counter reset is out before the units that count on it
          const unsigned tt = atomicAdd(&ctl->tail, (unsigned)upr);
          for (int i = 0; i < upr; ++i)
            if (tt + i < qcap)
              __hip_atomic_store(&q[tt + i], 1u + (((unsigned)m * 256u + (unsigned)(R + 1)) * 4u + (unsigned)i),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// the unit ring is cleared only when the previous solve queued something (behind the direct solver it hardly ever does)
__global__ void osjq_zero_kernel(unsigned* __restrict__ q, unsigned n, const OsjqCtl* __restrict__ ctl) {
  if (!ctl->dirty) return;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) q[i] = 0u;
}

// Behind the direct solver (one launch instead of five): the ring is cleared if the last solve used it, the control block
// reset, the matrices the certificate rejected queued -- with the trace / flag set-up the Jacobi needs for them
// (osj_begin_kernel's work, for those matrices only) -- and the queue sealed when there are none.  One workgroup.
__global__ __launch_bounds__(1024) void osjq_prep_direct_kernel(OsjqCtl* __restrict__ ctl, unsigned* __restrict__ q, unsigned qcap,
                                                               int* __restrict__ qdone, int* __restrict__ qclean, int batch,
                                                               int upr, const int* __restrict__ rep, const int* __restrict__ skip,
                                                               const double* __restrict__ Gc, int ldn, int nflag,
                                                               double* __restrict__ trace, int* __restrict__ notconv) {
  __shared__ int s_dirty;
  const int t = threadIdx.x;
  if (t == 0) s_dirty = ctl->dirty;
  __syncthreads();
  if (s_dirty)
    for (unsigned i = t; i < qcap; i += 1024) q[i] = 0u;
  __syncthreads();
  if (t == 0) {
    ctl->head = 0; ctl->tail = 0; ctl->finished = 0; ctl->nmat = 0; ctl->all_done = 0; ctl->error = 0; ctl->dirty = 0;
  }
  __syncthreads();
  for (int m = t; m < batch; m += 1024) {
    qdone[m] = 0;
    qclean[m] = 0;
    if (rep && rep[m] != m) continue;
    if (skip && skip[m]) continue;  // solved by the direct solver
    if (trace) {
      const double* M = Gc + (long)m * ldn * ldn;
      double sacc = 0.0;
      for (int i = 0; i < ldn; ++i) sacc += fabs(M[(long)i * ldn + i]);
      trace[m] = sacc;
      for (int f = 0; f < nflag; ++f) notconv[(long)f * batch + m] = 0;
    }
    const unsigned pos = atomicAdd(&ctl->tail, (unsigned)upr);
    for (int i = 0; i < upr; ++i) q[pos + i] = 1u + (((unsigned)m * 256u + 0u) * 4u + (unsigned)i);
    atomicAdd(&ctl->nmat, 1);
    ctl->dirty = 1;
  }
  __syncthreads();
  if (t == 0 && ctl->nmat == 0) ctl->all_done = 1;  // nothing queued: the consumers leave at once
}

template <int RP>
static void osjq_launch(EigPlan* p, hipStream_t st) {
  const int nb = p->ldn / OSJ_CB, upr = nb / 2;
  if (p->trd) {
    hipLaunchKernelGGL(osjq_prep_direct_kernel, dim3(1), dim3(1024), 0, st, p->qctl, p->q, p->qcap, p->qdone, p->qclean, p->batch,
                       upr, p->rep, p->trd_done, p->Gc, p->ldn, p->notconv ? p->sweeps * p->rps : 0, p->notconv ? p->trace : nullptr,
                       p->notconv);
  } else {
    hipLaunchKernelGGL(osjq_zero_kernel, dim3(cdiv(p->qcap, 1024)), dim3(1024), 0, st, p->q, p->qcap, p->qctl);
    hipLaunchKernelGGL(osjq_reset_kernel, dim3(1), dim3(1), 0, st, p->qctl);
    hipLaunchKernelGGL(osjq_init_kernel, dim3(cdiv(p->batch, 256)), dim3(256), 0, st, p->qctl, p->q, p->qdone, p->qclean,
                       p->batch, upr, p->rep, (const int*)nullptr);
  }
  long units = (long)p->batch * upr;
  // 2 resident workgroups per CU; fewer than that is fine too (nobody waits for a particular workgroup to be resident).
  // Behind the direct solver the queue only ever holds the few matrices its certificate rejected: a small grid then, so that
  // the usual launch -- which finds nothing to do -- does not make 512 workgroups wait for 64 KB of LDS each
  const long cap = p->trd ? 64 : 512;
  const int grid = (int)(units < cap ? units : cap);
  hipLaunchKernelGGL((osjq_kernel<RP>), dim3(grid), dim3(256), 0, st, p->Gc, p->ldn, nb, p->batch, p->sweeps, p->sort_from,
                     p->notconv, p->trace, p->q, p->qcap, p->qctl, p->qdone, p->qclean, p->q_timeout, p->err_out);
}

// orders <= 256: round 0 carries the pairs inside the blocks, nb - 1 launches per sweep
template <int RP>
static void osjw_launch_sweep(EigPlan* p, int sweep, hipStream_t st) {
  const int nb = p->ldn / OSJ_CB;
  const int so = sweep >= p->sort_from ? 1 : 0;
  const int rps = p->rps;  // nb - 1
  hipLaunchKernelGGL((osjw_kernel<RP, 8, true>), dim3(nb / 2, p->batch), dim3(256), 0, st, p->Gc, p->ldn, nb, 0,
                     p->notconv, sweep * rps, rps, p->trace, so, p->jrep);
  for (int round = 1; round < nb - 1; ++round)
    hipLaunchKernelGGL((osjw_kernel<RP, 8, false>), dim3(nb / 2, p->batch), dim3(256), 0, st, p->Gc, p->ldn, nb,
                       round, p->notconv, sweep * rps + round, rps, p->trace, so, p->jrep);
}

// G (batch x n x n, symmetric, row-major == column-major) -> Gc (batch x ldn x ldn), zero padded
__global__ void osj_pack_kernel(const double* __restrict__ G, int n, int ldn, double* __restrict__ Gc) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)ldn * ldn;
  if (gid >= per * gridDim.y) return;
  const int b = blockIdx.y;
  const long e = gid;
  if (e >= per) return;
  const int c = (int)(e / ldn), r = (int)(e - (long)c * ldn);
  Gc[b * per + e] = (c < n && r < n) ? G[(long)b * n * n + (long)c * n + r] : 0.0;
}

// Adaptive sweeps, start of a solve: trace[b] = sum |diag| of matrix b (scale of the convergence floor), flags
// of all sweeps cleared.  One wave per matrix.
__global__ void osj_begin_kernel(const double* __restrict__ Gc, int ldn, int nflag, double* __restrict__ trace,
                                 int* __restrict__ notconv) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const double* M = Gc + (long)b * ldn * ldn;
  double s = 0.0;
  for (int i = lane; i < ldn; i += 64) s += fabs(M[(long)i * ldn + i]);
  s = wave_sum(s);
  if (lane == 0) trace[b] = s;
  for (int f = lane; f < nflag; f += 64) notconv[(long)f * gridDim.x + b] = 0;
}

// Profiling only: adds the number of (matrix, launch round) pairs that did work in the solve just finished.
__global__ void osj_count_kernel(const int* __restrict__ notconv, int batch, int nround, int rps,
                                 unsigned long long* __restrict__ acc, const int* __restrict__ rep) {
  // (matrix, launch round) pairs that did work: a round runs unless the rps rounds before it were all clean
  int c = 0;
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    if (rep && rep[b] != b) continue;
    int clean = 0;
    for (int r = 0; r < nround; ++r) {
      if (r >= rps && clean >= rps) break;
      ++c;
      clean = notconv[(long)r * batch + b] ? 0 : clean + 1;
    }
  }
  c = (int)wave_sum((double)c);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, (unsigned long long)c);
}

// lam[b][j] = |column j| ; one wave per column
__global__ void osj_norms_kernel(const double* __restrict__ Gc, int ldn, double* __restrict__ lam, const int* __restrict__ skip) {
  const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (col >= ldn) return;
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  if (skip && skip[b]) return;  // solved by a direct solver: it wrote the eigenvalues itself
  const double* c = Gc + ((long)b * ldn + col) * ldn;
  double s = 0.0;
  for (int r = lane; r < ldn; r += 64) s += c[r] * c[r];
  s = wave_sum(s);
  if (lane == 0) lam[(long)b * ldn + col] = sqrt(s);
}

// evals (batch x n) and V (batch x n x n row-major, column j = eigenvector j)
__global__ void osj_extract_kernel(const double* __restrict__ Gc, const double* __restrict__ lam, int n, int ldn,
                                   double* __restrict__ evals, double* __restrict__ V) {
  const int b = blockIdx.y;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * n) return;
  const int j = (int)(gid / n), a = (int)(gid - (long)j * n);  // consecutive threads: consecutive rows of column j
  const double l = lam[(long)b * ldn + j];
  if (a == 0) evals[(long)b * n + j] = l;
  if (V) V[(long)b * n * n + (long)a * n + j] = (l > 0.0) ? Gc[((long)b * ldn + j) * ldn + a] / l : 0.0;
}

// orders 320-512: pairs inside the blocks by the row-per-thread kernel (NT = ldn threads), then all nb - 1
// block-pair rounds by the wave-private kernel with 4-column sets (8 waves)
template <int NT>
static void osjw4_launch_sweep(EigPlan* p, int sweep, hipStream_t st) {
  constexpr int RP = NT / 64;
  const int nb = p->ldn / OSJ_CB;
  const int rps = p->rps;  // nb: the intra-block launch + nb - 1 block-pair rounds
  hipLaunchKernelGGL((osj_round_kernel<OSJ_CB / 2, NT, 2>), dim3(nb, p->batch), dim3(NT), 0, st, p->Gc, p->n, p->ldn,
                     2 * nb, 0, p->notconv, sweep * rps, rps, p->trace, p->sortcols, p->jrep);
  const int so = sweep >= p->sort_from ? 1 : 0;
  for (int round = 0; round < nb - 1; ++round)
    hipLaunchKernelGGL((osjw_kernel<RP, 4, false>), dim3(nb / 2, p->batch), dim3(512), 0, st, p->Gc, p->ldn, nb, round,
                       p->notconv, sweep * rps + 1 + round, rps, p->trace, so, p->jrep);
}

template <int NT>
static void osj_launch_sweep(EigPlan* p, int sweep, hipStream_t st) {
  // one sweep = every column pair exactly once: one INTRA launch (pairs inside each group of 2*CB
  // columns... i.e. inside each block pair (2b, 2b+1)) followed by the block-pair rounds; the pair
  // (2b, 2b+1) itself meets in the round-robin too, where only its CROSS pairs are left to do.
  const int nb = p->ldn / OSJ_CB;
  // INTRA: pairs within each single block.  Run as MODE 2 on "super blocks" of 2*CB columns would also
  // rotate the cross pairs of (2b, 2b+1); instead launch MODE 2 with half-size blocks: CB/2 columns per
  // block -> 2*(CB/2) = CB columns per workgroup = exactly one block.
  const int rps = p->rps;  // nb
  hipLaunchKernelGGL((osj_round_kernel<OSJ_CB / 2, NT, 2>), dim3(nb, p->batch), dim3(NT), 0, st, p->Gc, p->n, p->ldn,
                     2 * nb, 0, p->notconv, sweep * rps, rps, p->trace, p->sortcols, p->jrep);
  for (int round = 0; round < nb - 1; ++round)
    hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, NT, 1>), dim3(nb / 2, p->batch), dim3(NT), 0, st, p->Gc, p->n, p->ldn,
                       nb, round, p->notconv, sweep * rps + 1 + round, rps, p->trace, p->sortcols, p->jrep);
}

static int osj_enqueue_sweeps(EigPlan* p, hipStream_t st) {
  if (p->use_queue) {
    switch (p->ldn) {
      case 64: osjq_launch<1>(p, st); break;
      case 128: osjq_launch<2>(p, st); break;
      case 192: osjq_launch<3>(p, st); break;
      default: osjq_launch<4>(p, st); break;
    }
    MUSED_LAUNCH_CHECK();
    return MUSED_OK;
  }
  for (int sw = 0; sw < p->sweeps; ++sw) {
    if (p->wavek) {
      switch (p->ldn) {
        case 64: osjw_launch_sweep<1>(p, sw, st); break;
        case 128: osjw_launch_sweep<2>(p, sw, st); break;
        case 192: osjw_launch_sweep<3>(p, sw, st); break;
        case 256: osjw_launch_sweep<4>(p, sw, st); break;
        case 320: osjw4_launch_sweep<320>(p, sw, st); break;
        case 384: osjw4_launch_sweep<384>(p, sw, st); break;
        case 448: osjw4_launch_sweep<448>(p, sw, st); break;
        default: osjw4_launch_sweep<512>(p, sw, st); break;
      }
      continue;
    }
    switch (p->ldn) {
      case 64: osj_launch_sweep<64>(p, sw, st); break;
      case 128: osj_launch_sweep<128>(p, sw, st); break;
      case 192: osj_launch_sweep<192>(p, sw, st); break;
      case 256: osj_launch_sweep<256>(p, sw, st); break;
      case 320: osj_launch_sweep<320>(p, sw, st); break;
      case 384: osj_launch_sweep<384>(p, sw, st); break;
      case 448: osj_launch_sweep<448>(p, sw, st); break;
      case 512: osj_launch_sweep<512>(p, sw, st); break;
      case 640: osj_launch_sweep<640>(p, sw, st); break;
      case 768: osj_launch_sweep<768>(p, sw, st); break;
      case 896: osj_launch_sweep<896>(p, sw, st); break;
      case 1024: osj_launch_sweep<1024>(p, sw, st); break;
      default: set_error("osj: unsupported padded order %d", p->ldn); return MUSED_ERR_UNSUPPORTED;
    }
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

static int osj_padded_order(int n) {
  if (n <= 512) return ((n + 63) / 64) * 64;
  return ((n + 127) / 128) * 128;
}

int eig_plan_create(int n, int batch, int sweeps, bool own_graph, EigPlan** out, const int* rep, int flags, int* err_out,
                    int need) {
  MUSED_REQUIRE(n >= 2 && n % 2 == 0 && n <= 1024 && batch >= 1 && sweeps >= 1,
                "eig_plan_create: the order must be even and <= 1024 (n=%d)", n);
  CaptureLock resource_guard(capture_mutex());  // allocations + capture: not beside another thread's capture
  EigPlan* p = new EigPlan();
  memset(p, 0, sizeof(*p));
  p->n = n; p->batch = batch; p->sweeps = sweeps;
  p->rep = rep;
  p->jrep = rep;
  p->err_out = err_out;
  p->method = 1;
  const size_t bytes = sizeof(double) * (size_t)batch * n * n;
  if (p->method == 1) {
    p->ldn = osj_padded_order(n);
    MUSED_CHECK_HIP(hipMalloc(&p->G[0], bytes));
    MUSED_CHECK_HIP(hipMalloc(&p->Gc, sizeof(double) * (size_t)batch * p->ldn * p->ldn));
    MUSED_CHECK_HIP(hipMalloc(&p->lam, sizeof(double) * (size_t)batch * p->ldn));
    const char* sf = getenv("MUSED_OSJ_SORT_FROM");
    p->sort_from = sf ? atoi(sf) : 0;
    const char* wk = getenv("MUSED_OSJ_WAVE");  // 0: row-per-thread kernel for every order
    p->wavek = (p->ldn <= 512 && !(wk && wk[0] == '0')) ? 1 : 0;
    const char* so = getenv("MUSED_OSJ_SORT");
    p->sortcols = (so && so[0] == '0') ? 0 : 1;  // row-per-thread kernel only: helps on rank-deficient matrices
    if (flags & EIG_PLAN_NO_SORT) {  // columns stay where they are (the caller reads parts of them by position)
      p->sort_from = 1 << 30;
      p->sortcols = 0;
    }
    {
      const int nb0 = p->ldn / OSJ_CB;
      p->rps = (p->wavek && p->ldn <= 256) ? (nb0 > 1 ? nb0 - 1 : 1) : nb0;
    }
    {
      // persistent work-queue solver: wave-private kernel orders (<= 256), at most 255 global rounds per solve
      // Default (MUSED_EIG_QUEUE unset): batches of at most 224 units per round (one sketch lane: 28 matrices x 4) --
      // the persistent workgroups then occupy at most one slot on fewer than all CUs, so kernels of other streams
      // (the adjacency / eigenstep chain, whose one-workgroup panel kernels need a whole CU) still find room.  Larger
      // batches keep the launch-per-round graph: a persistent grid that fills the GPU for milliseconds would starve them.
      // MUSED_EIG_QUEUE=1 / 0 forces it on / off.
      const char* qe = getenv("MUSED_EIG_QUEUE");
      const int nbq = p->ldn / OSJ_CB;
      const bool fits = p->wavek && p->ldn <= 256 && nbq >= 2 && sweeps * (nbq - 1) <= 255 && batch <= (1 << 21);
      bool want = qe ? (qe[0] == '1') : ((long)batch * (nbq / 2) <= 224);
      {
        // direct solver (MUSED_EIG_TRD=0 turns it off): orders up to 256, callers that read the leading pairs only -- the
        // top half (FD rotation: those that survive the shrink are certified) or the `need` largest, all certified (eigenstep)
        const char* te = getenv("MUSED_EIG_TRD");
        p->trd_need = (flags & (EIG_PLAN_TOP_NEED | EIG_PLAN_TOP_FD)) ? need : ((flags & EIG_PLAN_TOP_HALF) ? n / 2 : 0);
        p->trd_cert_all = (flags & EIG_PLAN_TOP_NEED) ? 1 : 0;
        p->trd = (p->trd_need > 0 && trd_supports(n, p->ldn, p->trd_need) && fits && !(te && te[0] == '0')) ? 1 : 0;
        if (p->trd) want = true;  // its rare rejects go through the one-launch queue solver (nothing queued: it exits at once)
      }
      p->use_queue = (fits && want) ? 1 : 0;
      if (p->use_queue) {
        p->qcap = (unsigned)((long)batch * (nbq / 2) * (nbq - 1) * sweeps);
        MUSED_CHECK_HIP(hipMalloc(&p->q, sizeof(unsigned) * (size_t)p->qcap));
        MUSED_CHECK_HIP(hipMalloc(&p->qctl, sizeof(OsjqCtl)));
        MUSED_CHECK_HIP(hipMalloc(&p->qdone, sizeof(int) * (size_t)batch));
        MUSED_CHECK_HIP(hipMalloc(&p->qclean, sizeof(int) * (size_t)batch));
        MUSED_CHECK_HIP(hipMemset(p->qctl, 0, sizeof(OsjqCtl)));
        MUSED_CHECK_HIP(hipMemset(p->q, 0, sizeof(unsigned) * (size_t)p->qcap));  // (cleared from now on only after use)
        const char* tq = getenv("MUSED_EIG_QUEUE_TIMEOUT_TICKS");  // test knob: 1 forces the give-up path
        p->q_timeout = tq ? strtoull(tq, nullptr, 10) : 300000000ull;
      }
      if (p->trd) {
        int rc_t = trd_prepare();
        if (rc_t) return rc_t;
        MUSED_CHECK_HIP(hipMalloc(&p->trd_ws, sizeof(double) * trd_workspace_doubles(batch)));
        MUSED_CHECK_HIP(hipMalloc(&p->trd_done, sizeof(int) * (size_t)batch));
      } else if (p->trd_need > 0 && trdx_supports(p->ldn, p->trd_need) &&
                 !(getenv("MUSED_EIG_TRD") && getenv("MUSED_EIG_TRD")[0] == '0')) {
        // orders 320 .. 1024: blocked direct solver; the sweep graph below runs on the matrices it rejects (jrep)
        p->trd = 2;
        int rc_t = trdx_prepare(p->ldn);
        if (rc_t) return rc_t;
        MUSED_CHECK_HIP(hipMalloc(&p->trd_ws, sizeof(double) * trdx_workspace_doubles(p->ldn, batch)));
        MUSED_CHECK_HIP(hipMalloc(&p->trd_done, sizeof(int) * (size_t)batch));
        MUSED_CHECK_HIP(hipMalloc(&p->trdx_act, sizeof(int) * (size_t)batch));
        MUSED_CHECK_HIP(hipMalloc(&p->trdx_jrep, sizeof(int) * (size_t)batch));
        MUSED_CHECK_HIP(hipMalloc(&p->trdx_nrej, sizeof(int)));
        MUSED_CHECK_HIP(hipHostMalloc((void**)&p->trdx_nrej_host, sizeof(int), hipHostMallocDefault));
        MUSED_CHECK_HIP(hipEventCreateWithFlags(&p->trdx_ev, hipEventDisableTiming));
        p->jrep = p->trdx_jrep;
      }
    }
    // Adaptive sweep count (default; MUSED_EIG_ADAPTIVE=0: always `sweeps` sweeps): `sweeps` is the cap, a matrix
    // stops after the first sweep that met no column pair with cos^2 above the threshold (osj_pair_active).  How many sweeps
    // that takes depends on the matrix (full-rank sketch buffers ~10 at order 256, rank-deficient ones up to 16).
    const char* ad = getenv("MUSED_EIG_ADAPTIVE");
    if (!(ad && ad[0] == '0') && !(flags & EIG_PLAN_FIXED_SWEEPS)) {
      // [trace (batch doubles) | work counter (1 x u64) | flags (batch * sweeps ints)]
      MUSED_CHECK_HIP(hipMalloc(&p->trace, sizeof(double) * ((size_t)batch + 1) + sizeof(int) * (size_t)batch * sweeps * p->rps));
      p->work = (unsigned long long*)(p->trace + batch);
      p->notconv = (int*)(p->trace + batch + 1);
      MUSED_CHECK_HIP(hipMemset(p->work, 0, 8));
    }
  }
  p->have_graph = false;
  const char* ng = getenv("MUSED_NO_GRAPH");
  if (own_graph && !(ng && ng[0] == '1')) {
    MUSED_CHECK_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
    MUSED_CHECK_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = osj_enqueue_sweeps(p, p->cap_stream);
    hipError_t e = hipStreamEndCapture(p->cap_stream, &p->graph);
    if (rc < 0 || e != hipSuccess) {
      set_error("eig_plan_create: graph capture failed (%s)", hipGetErrorString(e));
      return MUSED_ERR_HIP;
    }
    MUSED_CHECK_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    p->have_graph = true;
    // the capture stream has done its job: every live HIP stream competes for the few hardware queues the user's
    // streams are mapped onto (two sketch groups + the main path need three of them undisturbed)
    (void)hipStreamDestroy(p->cap_stream);
    p->cap_stream = nullptr;
  }
  *out = p;
  return MUSED_OK;
}

void eig_plan_destroy(EigPlan* p) {
  if (!p) return;
  CaptureLock resource_guard(capture_mutex());
  if (p->have_graph) {
    (void)hipGraphExecDestroy(p->exec);
    (void)hipGraphDestroy(p->graph);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  }
  for (int i = 0; i < 2; ++i) {
    if (p->G[i]) (void)hipFree(p->G[i]);
    if (p->V[i]) (void)hipFree(p->V[i]);
  }
  if (p->Gc) (void)hipFree(p->Gc);
  if (p->lam) (void)hipFree(p->lam);
  if (p->trace) (void)hipFree(p->trace);
  if (p->q) (void)hipFree(p->q);
  if (p->qctl) (void)hipFree(p->qctl);
  if (p->qdone) (void)hipFree(p->qdone);
  if (p->qclean) (void)hipFree(p->qclean);
  if (p->trd_ws) (void)hipFree(p->trd_ws);
  if (p->trd_done) (void)hipFree(p->trd_done);
  if (p->trdx_act) (void)hipFree(p->trdx_act);
  if (p->trdx_jrep) (void)hipFree(p->trdx_jrep);
  if (p->trdx_nrej) (void)hipFree(p->trdx_nrej);
  if (p->trdx_nrej_host) (void)hipHostFree(p->trdx_nrej_host);
  if (p->trdx_ev) (void)hipEventDestroy(p->trdx_ev);
  if (p->ev0) {
    for (size_t i = 0; i < p->ev0->size(); ++i) {
      (void)hipEventDestroy((*p->ev0)[i]);
      (void)hipEventDestroy((*p->ev1)[i]);
      if (p->evm) (void)hipEventDestroy((*p->evm)[i]);
    }
    delete p->ev0;
    delete p->ev1;
    delete p->evm;
  }
  delete p;
}


// Live timing of the Jacobi sweeps: between enable and read every replay of the sweep graph is bracketed
// by HIP events on the launch stream.  Read (blocking) returns the summed duration, the number of
// osj_round_kernel launches it covers and the bytes one launch moves (every matrix element read once
// and written once).
int eig_plan_profile(EigPlan* p, bool on) {
  if (!p || p->method != 1) return MUSED_OK;
  if (on && !p->ev0) {
    p->ev0 = new std::vector<hipEvent_t>(2048);
    p->ev1 = new std::vector<hipEvent_t>(2048);
    if (p->trd) p->evm = new std::vector<hipEvent_t>(2048);
    for (size_t i = 0; i < p->ev0->size(); ++i) {
      MUSED_CHECK_HIP(hipEventCreate(&(*p->ev0)[i]));
      MUSED_CHECK_HIP(hipEventCreate(&(*p->ev1)[i]));
      if (p->evm) MUSED_CHECK_HIP(hipEventCreate(&(*p->evm)[i]));
    }
  }
  p->prof = on;
  if (on) {
    p->prof_n = 0;
    if (p->work) MUSED_CHECK_HIP(hipMemset(p->work, 0, 8));
  }
  return MUSED_OK;
}

int eig_plan_profile_read(EigPlan* p, double* total_ms, long* launches, double* bytes_per_launch) {
  *total_ms = 0.0; *launches = 0; *bytes_per_launch = 0.0;
  if (!p || p->method != 1 || !p->ev0) return MUSED_OK;
  for (int i = 0; i < p->prof_n; ++i) {
    MUSED_CHECK_HIP(hipEventSynchronize((*p->ev1)[i]));
    float ms = 0.f;
    MUSED_CHECK_HIP(hipEventElapsedTime(&ms, (*p->ev0)[i], (*p->ev1)[i]));
    *total_ms += ms;
  }
  const int nb = p->ldn / OSJ_CB;
  // per sweep: (nb - 1) block-pair rounds, plus one launch for the pairs inside the blocks in the row-per-thread kernel
  *launches = (long)p->prof_n * p->sweeps * p->rps;
  (void)nb;
  // a launch reads and writes every element of every matrix that is still iterating; with the adaptive sweep
  // count the launches of later sweeps find fewer (or no) such matrices: average over the launches timed
  double frac = 1.0;
  if (p->work && p->prof_n > 0) {
    unsigned long long w = 0;
    CaptureLock guard(capture_mutex());  // (a blocking legacy-stream copy: not beside another thread's stream capture)
    MUSED_CHECK_HIP(hipMemcpy(&w, p->work, 8, hipMemcpyDeviceToHost));
    frac = (double)w / ((double)p->prof_n * p->sweeps * p->rps * p->batch);
  }
  *bytes_per_launch = 16.0 * (double)p->batch * p->ldn * p->ldn * frac;
  return MUSED_OK;
}

int eig_plan_profile_read_direct(EigPlan* p, double* total_ms, long* launches, double* matrices_solved, double* tridiag_ms) {
  *total_ms = 0.0; *launches = 0; *matrices_solved = 0.0;
  if (tridiag_ms) *tridiag_ms = 0.0;
  if (!p || !p->trd || !p->ev0) return MUSED_OK;
  for (int i = 0; i < p->prof_n; ++i) {
    MUSED_CHECK_HIP(hipEventSynchronize((*p->ev1)[i]));
    float ms = 0.f;
    MUSED_CHECK_HIP(hipEventElapsedTime(&ms, (*p->ev0)[i], (*p->ev1)[i]));
    *total_ms += ms;
    if (tridiag_ms && p->evm) {
      MUSED_CHECK_HIP(hipEventElapsedTime(&ms, (*p->ev0)[i], (*p->evm)[i]));
      *tridiag_ms += ms;
    }
  }
  *launches = p->prof_n;
  if (p->work) {
    unsigned long long w = 0;
    CaptureLock guard(capture_mutex());  // (a blocking legacy-stream copy: not beside another thread's stream capture)
    MUSED_CHECK_HIP(hipMemcpy(&w, p->work, 8, hipMemcpyDeviceToHost));
    *matrices_solved = (double)w;
  }
  return MUSED_OK;
}

// Column-form access for callers that can use the raw result of the one-sided solver: after
// eig_plan_run_inplace(p, nullptr, nullptr, ...) column j of matrix b, cols[(b * ld + j) * ld + 0..n), is
// lam_j u_j and lam[b * ld + j] its norm (the eigenvalue).  Returns false for the two-sided fallback.
bool eig_plan_columns(EigPlan* p, const double** cols, const double** lam, int* ld) {
  const char* rw = getenv("MUSED_EIG_RAW");
  if (p->method != 1 || (rw && rw[0] == '0')) return false;
  *cols = p->Gc;
  *lam = p->lam;
  *ld = p->ldn;
  return true;
}

bool eig_plan_direct_solver(EigPlan* p) { return p && p->trd != 0; }

// Input buffer for a caller that writes the matrices itself; when the order needs no padding this is the
// solver's working copy (no pack pass).
double* eig_plan_input(EigPlan* p) {
  const char* dd = getenv("MUSED_EIG_DIRECT");
  if (p->method == 1 && p->n == p->ldn && !(dd && dd[0] == '0')) {
    p->direct = true;
    return p->Gc;
  }
  return p->G[0];
}

int eig_plan_run_inplace(EigPlan* p, double* evals, double* V, hipStream_t st, bool allow_graph) {
  if (p->method == 1) {
    const long per = (long)p->ldn * p->ldn;
    if (!p->direct)
      hipLaunchKernelGGL(osj_pack_kernel, dim3(cdiv(per, 256), p->batch), dim3(256), 0, st, p->G[0], p->n, p->ldn, p->Gc);
    if (p->notconv && !(p->trd == 1))  // (behind the register-resident direct solver the queue's set-up kernel does it, for the
      hipLaunchKernelGGL(osj_begin_kernel, dim3(p->batch), dim3(64), 0, st, p->Gc, p->ldn, p->sweeps * p->rps, p->trace,   // rejected matrices only)
                         p->notconv);
    const bool rec = p->prof && p->ev0 && p->prof_n < (int)p->ev0->size();
    if (rec) MUSED_CHECK_HIP(hipEventRecord((*p->ev0)[p->prof_n], st));
    if (p->trd) {
      if (p->trd == 2) MUSED_CHECK_HIP(hipMemsetAsync(p->trdx_nrej, 0, sizeof(int), st));
      const int rc = p->trd == 2
                         ? trdx_solve(p->Gc, p->ldn, p->trd_need, p->trd_cert_all != 0, p->batch, p->rep, p->trd_done, p->trdx_act,
                                      p->trdx_jrep, p->trdx_nrej, p->trd_ws, st, rec ? p->work : nullptr,
                                      (rec && p->evm) ? (*p->evm)[p->prof_n] : nullptr, nullptr, p->lam)
                         : trd_solve(p->Gc, p->n, p->ldn, p->trd_need, p->trd_cert_all != 0, p->batch, p->rep, p->trd_done,
                                     p->trd_ws, st, nullptr, rec ? p->work : nullptr,
                                     (rec && p->evm) ? (*p->evm)[p->prof_n] : nullptr, p->lam);
      if (rc) return rc;
      if (rec) {  // the events of a direct-solver plan bracket the direct solver alone (the Jacobi behind it only sees rejects)
        MUSED_CHECK_HIP(hipEventRecord((*p->ev1)[p->prof_n], st));
        ++p->prof_n;
      }
    }
    bool run_jacobi = true;
    if (p->trd == 2) {
      // The blocked direct solver's fallback is the sweep graph: sweeps x rounds launches over the whole batch, each of which
      // only finds out on the device that it has nothing to do (12 % of config 3's kernel time when it was launched
      // unconditionally).  A rejection is rare, a solve of these orders takes milliseconds: the host waits for the count
      // (an event on this stream: safe beside other threads' captures) and launches the graph only when it is not zero.
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cs);
      if (cs == hipStreamCaptureStatusNone) {
        MUSED_CHECK_HIP(hipMemcpyAsync(p->trdx_nrej_host, p->trdx_nrej, sizeof(int), hipMemcpyDeviceToHost, st));
        MUSED_CHECK_HIP(hipEventRecord(p->trdx_ev, st));
        MUSED_CHECK_HIP(hipEventSynchronize(p->trdx_ev));
        run_jacobi = *p->trdx_nrej_host != 0;
      }
    }
    if (!run_jacobi) {
    } else if (p->have_graph && allow_graph) {
      MUSED_CHECK_HIP(hipGraphLaunch(p->exec, st));
    } else {
      const int rc = osj_enqueue_sweeps(p, st);
      if (rc) return rc;
    }
    if (rec && !p->trd) {
      MUSED_CHECK_HIP(hipEventRecord((*p->ev1)[p->prof_n], st));
      ++p->prof_n;
      if (p->notconv)
        hipLaunchKernelGGL(osj_count_kernel, dim3(1), dim3(256), 0, st, p->notconv, p->batch, p->sweeps * p->rps, p->rps,
                           p->work, p->rep);
    }
    hipLaunchKernelGGL(osj_norms_kernel, dim3(cdiv(p->ldn, 4), p->batch), dim3(256), 0, st, p->Gc, p->ldn, p->lam,
                       p->trd ? p->trd_done : (const int*)nullptr);
    if (evals)
      hipLaunchKernelGGL(osj_extract_kernel, dim3(cdiv((long)p->n * p->n, 256), p->batch), dim3(256), 0, st, p->Gc, p->lam,
                         p->n, p->ldn, evals, V);
    MUSED_LAUNCH_CHECK();
    return MUSED_OK;
  }
  set_error("eig_plan_run_inplace: unsupported solver");
  return MUSED_ERR_STATE;
}

int eig_plan_run(EigPlan* p, const double* G, double* evals, double* V, hipStream_t st) {
  const size_t bytes = sizeof(double) * (size_t)p->batch * p->n * p->n;
  MUSED_CHECK_HIP(hipMemcpyAsync(p->G[0], G, bytes, hipMemcpyDeviceToDevice, st));
  p->direct = false;
  return eig_plan_run_inplace(p, evals, V, st, true);
}

}  // namespace mused

using namespace mused;

extern "C" {

// Diagnostic: average kernel time (us, HIP events) of `reps` cross-round launches of the one-sided Jacobi
// kernel on `batch` matrices of order 256, for timing ablation `variant` (0 = the real kernel).
int mused_debug_osj_time(const double* init, int batch, int variant, int reps, double* out_us, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int ldn = 256, nb = ldn / OSJ_CB;
  double* G = nullptr;
  MUSED_CHECK_HIP(hipMalloc(&G, sizeof(double) * (size_t)batch * ldn * ldn));
  MUSED_CHECK_HIP(hipMemcpy(G, init, sizeof(double) * (size_t)batch * ldn * ldn, hipMemcpyDeviceToDevice));
  hipEvent_t e0, e1;
  MUSED_CHECK_HIP(hipEventCreate(&e0));
  MUSED_CHECK_HIP(hipEventCreate(&e1));
  auto launch = [&](int round) {
    dim3 grid(nb / 2, batch), blk(256);
    switch (variant) {
      case 5: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 5>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr); break;
      case 7: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 7>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr); break;
      case 2: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 2>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr); break;
      case 3: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 3>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr); break;
      case 4: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 4>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr); break;
      default: hipLaunchKernelGGL((osj_round_kernel<OSJ_CB, 256, 1, 0>), grid, blk, 0, st, G, 256, ldn, nb, round, (int*)nullptr, 0, 1, (const double*)nullptr, 1, (const int*)nullptr);
    }
  };
  MUSED_CHECK_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch(i % (nb - 1));
  MUSED_CHECK_HIP(hipEventRecord(e1, st));
  MUSED_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MUSED_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *out_us = 1e3 * ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(G);
  return MUSED_OK;
}

// Diagnostic (not part of the declared ABI): average time (ms, HIP events) of `reps` solves of `batch` matrices of order n
// with one plan (created under the current environment: MUSED_EIG_QUEUE etc.); evals / V of the last solve are returned.
int mused_debug_eig_time(const double* G, int n, int batch, int sweeps, int reps, double* evals, double* V, double* out_ms,
                         int* out_err, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  EigPlan* p = nullptr;
  int rc = eig_plan_create(n, batch, sweeps, true, &p);
  if (rc) return rc;
  hipEvent_t e0, e1;
  MUSED_CHECK_HIP(hipEventCreate(&e0));
  MUSED_CHECK_HIP(hipEventCreate(&e1));
  rc = eig_plan_run(p, G, evals, V, st);  // warm-up (graph upload)
  MUSED_CHECK_HIP(hipStreamSynchronize(st));
  MUSED_CHECK_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps && !rc; ++i) rc = eig_plan_run(p, G, evals, V, st);
  MUSED_CHECK_HIP(hipEventRecord(e1, st));
  MUSED_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MUSED_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *out_ms = ms / (reps > 0 ? reps : 1);
  if (out_err) {
    *out_err = 0;
    if (p->use_queue) {
      OsjqCtl c;
      MUSED_CHECK_HIP(hipMemcpy(&c, p->qctl, sizeof(c), hipMemcpyDeviceToHost));
      *out_err = c.error ? -1 : (int)c.tail;  // units queued in the last solve (or -1 on a timeout)
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  eig_plan_destroy(p);
  return rc;
}

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
