// Direct symmetric eigensolver for the 2l x 2l = 256 x 256 Gram matrix of an FD rotation (the "per-window SVD / rotation
// step" of SWFD, a6 of SURVEY section 8): ONE workgroup per matrix, the matrix resident in registers, ~2.4 n^3 flop
// instead of the ~31 n^3 of the one-sided Jacobi (eig.hip), which stays as the fallback and for every other order.
//
//   A  Householder tridiagonalisation  G = Q T Q^T          (LAPACK dsytd2, lower variant: 254 reflectors)
//   B  the 128 largest eigenvalues of T by multisection on Sturm counts (4 lanes per eigenvalue, 5 sub-intervals per pass)
//   C  their eigenvectors by twisted factorisation of T - lam I (Fernando 1997 / LAPACK dlar1v on T itself): forward and
//      backward pivot sequences, twist index k = argmin |gamma_k|, z_k = 1 and two two-term recurrences outwards
//   D  back-transformation  V = Q Z  (reflectors applied in reverse), columns written as lam_j v_j -- the form the
//      one-sided Jacobi leaves (eig_plan_columns), so the FD `decide` kernel reads either solver's output.
//
// Only the top l = 128 eigenpairs are formed: the FD shrink uses lam_0 .. lam_{l-1} (delta = lam_{l-1}) and the vectors of
// the directions whose shrunk energy survives.  CERTIFICATE instead of trust: a matrix is handed to the Jacobi solver
// (done[b] = 0, its input left untouched) when any significant vector has a twisted-factorisation residual above 1e-11 |T|,
// a non-finite norm, a cosine above 1e-8 with one of its 4 neighbours in the spectrum, or when 6 significant eigenvalues lie
// within 1e-7 lam_0 of each other (eigenvalue clusters: multiple eigenvalues make the vectors of a cluster non-orthogonal).
//
// Data layout of phase A (512 threads = a 16 x 32 grid (p, q), 8 waves = 2 x 4, lanes = 8 x 8): thread (p, q) holds the
// elements (i, j) with i = p + 16 u + 32 a, j = q + 32 b for u < 2, 0 <= b <= a < 8 -- a 2-D cyclic distribution of the
// lower block triangle, 72 doubles per thread, so the active trailing matrix shrinks evenly over the threads.  The diagonal
// blocks (a == b) hold BOTH triangles at HALF weight: then  y = A v  is  y_i = sum_j H_ij v_j (row part) + sum_i H_ij v_i
// (column part) with one uniform expression for every stored element, and the rank-2 update needs no masks either.
#include <stdlib.h>

#include "trd_common.h"

namespace mused {

constexpr int TN = 256, TM = 128, TNT = 512;
// LDS map (doubles): persistent part, then a scratch region reused by the phases
constexpr int L_D = 0, L_E = 256, L_LAM = 1024, L_ZS = 1152, L_TAU = 1280, L_MISC = 1536, L_S = 1728;
// phase A scratch
constexpr int A_XS = 0 /* [256] */, A_VS = 256, A_WS = 512, A_RP = 768 /* [4][256] */, A_CP = 1792 /* [2][256] */, A_RED = 2304 /* [8] */, A_SQ = 2312 /* [2] */;

__host__ __device__ constexpr int tidx(int a, int b) { return a * (a + 1) / 2 + b; }

// ---- phase A: one Householder step (column k), K = k / 32 selects which register blocks are still active -------------------
// Four workgroup barriers per step: column k -> LDS | Householder vector v -> LDS | partial sums of y = A v (and of v . y)
// -> LDS | w -> LDS.  v and w are stored as plain vectors (zeros where the reflector does not act), so that the 2 x 36
// register elements of a thread are updated from unconditional LDS reads.
#ifdef TRD_STEP_PROFILE
#define TRD_TICK(i) do { const long long now_ = clock64(); prof[i] += now_ - last_; last_ = now_; } while (0)
#else
#define TRD_TICK(i) do { } while (0)
#endif
template <int K>
__device__ __forceinline__ void trd_step(double (&A)[2][36], const int k, double* __restrict__ sm, const int t, const int p,
                                         const int q, const int wp, const int wq, double* __restrict__ Hs, long long* prof) {
  const int kk = k & 31, l = t & 63;
#ifdef TRD_STEP_PROFILE
  long long last_ = clock64();
#endif
  double* S = sm + L_S;
  double* xs = S + A_XS;
  double* vs = S + A_VS;
  double* ws = S + A_WS;
  // (1) column k below the diagonal -> xs (the 16 threads of grid column q == kk hold it: 8 adjacent lanes in each of two
  //     waves), and the sum of squares below row k + 1 from their registers: 8-lane butterfly, one partial per wave
  if (q == kk) {
    double sqp = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int a = K; a < 8; ++a) {
        const int i = p + 16 * u + 32 * a;
        const double x = (a == K ? 2.0 : 1.0) * A[u][tidx(a, K)];
        xs[i] = x;
        if (a <= K + 1) sqp += (i > k + 1) ? x * x : 0.0;  // (row k + 1 lies in block K, or is the first row of block K + 1)
        else sqp = fma(x, x, sqp);
      }
    sqp += dpp_mov_f64<DPP_QUAD_XOR1>(sqp);
    sqp += dpp_mov_f64<DPP_QUAD_XOR2>(sqp);
    sqp += dpp_mov_f64<DPP_ROW_HALF_MIRROR>(sqp);
    if ((l & 7) == 0) S[A_SQ + wp] = sqp;
    if (p == (k & 15)) sm[L_D + k] = 2.0 * ((kk >> 4) ? A[1][tidx(K, K)] : A[0][tidx(K, K)]);
  }
  lds_barrier();
  TRD_TICK(0);
  // (2) Householder vector (dlarfg), every thread for itself from the two partial sums:
  //     beta = -sign(x0) |x|, tau = (beta - x0) / beta, v = x / (x0 - beta)
  const double sq = S[A_SQ] + S[A_SQ + 1];
  const double x0 = xs[k + 1];
  double tau = 0.0, beta = x0, scale = 0.0;
  // (a sum of squares in the denormal range -- the rounding residue of an exactly rank-deficient matrix can deflate level by
  // level down to there: every row the same -- is a zero column: rsq / rcp have no Newton step that survives it.  LAPACK's
  // dlarfg rescales instead; entries below 1e-140 are nothing a Gram matrix of this path resolves.)
  if (sq > 1e-280) {
    const double h = fma(x0, x0, sq);
    double rs = __builtin_amdgcn_rsq(h);  // 1 / sqrt(h): seed + two Newton steps
    rs = rs * fma(-0.5 * h, rs * rs, 1.5);
    rs = rs * fma(-0.5 * h, rs * rs, 1.5);
    const double nrm = h * rs;
    beta = x0 >= 0.0 ? -nrm : nrm;
    tau = (beta - x0) * trd_rcp(beta);
    scale = trd_rcp(x0 - beta);
  }
  if (t < TN) {
    const double v = (tau != 0.0) ? (t > k + 1 ? xs[t] * scale : (t == k + 1 ? 1.0 : 0.0)) : 0.0;
    vs[t] = v;
    Hs[(long)k * TN + t] = v;
    if (t == 0) {
      sm[L_E + k] = beta;
      sm[L_TAU + k] = tau;
    }
  }
  lds_barrier();
  TRD_TICK(1);
  if (tau == 0.0) return;  // H = I (uniform over the workgroup: every thread computed the same scalars from the same data)
  // (3) y = A v over the stored elements: row part r (to be summed over q) and column part c (to be summed over p).
  //     v . y = v^T A v = 2 sum over the stored elements of H_ij v_i v_j = 2 sum_threads sum_rows v_i r_i: no second pass.
  {
    double r[16];  // r[2 a + u]: row i = p + 16 (2 a + u)
#pragma unroll
    for (int e = 0; e < 16; ++e) r[e] = 0.0;
#pragma unroll
    for (int b = K; b < 8; ++b) {
      const double vj = vs[q + 32 * b];
#pragma unroll
      for (int a = b; a < 8; ++a) {
        r[2 * a] = fma(A[0][tidx(a, b)], vj, r[2 * a]);
        r[2 * a + 1] = fma(A[1][tidx(a, b)], vj, r[2 * a + 1]);
      }
    }
    double c[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) c[b] = 0.0;
    double dpart = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int a = K; a < 8; ++a) {
        const double vi = vs[p + 16 * u + 32 * a];
        dpart = fma(vi, r[2 * a + u], dpart);
#pragma unroll
        for (int b = K; b <= a; ++b) c[b] = fma(A[u][tidx(a, b)], vi, c[b]);
      }
    // (4) in-wave transposing reductions: r (16 values) over the 8 lanes that differ in lq = lane bits 3-5 -- two stages of
    //     v_permlane{32,16}_swap (3 instructions per pair) and one row rotation --, c (8 values) over lp = lane bits 0-2 by DPP
    //     (7 instructions per pair: two selects, the move, the add).  From column 128 on (K >= 4) the rows and columns below
    //     128 are out of the trailing matrix: their sums are zeros and the first stage of either reduction is skipped.
    const bool h5 = (l & 32) != 0, h4 = (l & 16) != 0, h3 = (l & 8) != 0;
    const bool h2 = (l & 4) != 0, h1 = (l & 2) != 0, h0 = (l & 1) != 0;
    double* rp = S + A_RP + wq * 256;
    double* cpw = S + A_CP + wp * 256;
    if constexpr (K < 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = swap32_add(r[e], r[e + 8]);  // lanes 0-31 keep e, lanes 32-63 e + 8
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = swap16_add(r[e], r[e + 4]);  // bit 4 clear keeps e, set e + 4
#pragma unroll
      for (int e = 0; e < 2; ++e) {  // l ^ 8
        const double keep = h3 ? r[e + 2] : r[e], send = h3 ? r[e] : r[e + 2];
        r[e] = keep + dpp_mov_f64<DPP_ROW_ROR8>(send);
      }
      const int e0 = (h5 ? 8 : 0) + (h4 ? 4 : 0) + (h3 ? 2 : 0);  // this lane ends with r-values e0, e0 + 1
      rp[p + 16 * e0] = r[0];
      rp[p + 16 * (e0 + 1)] = r[1];
#pragma unroll
      for (int b = 0; b < 4; ++b) {  // partner 7 - (l & 7) of the 8-lane group: decided by bit 2
        const double keep = h2 ? c[b + 4] : c[b], send = h2 ? c[b] : c[b + 4];
        c[b] = keep + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(send);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {  // l ^ 1
        const double keep = h0 ? c[b + 2] : c[b], send = h0 ? c[b] : c[b + 2];
        c[b] = keep + dpp_mov_f64<DPP_QUAD_XOR1>(send);
      }
      {  // l ^ 2
        const double keep = h1 ? c[1] : c[0], send = h1 ? c[0] : c[1];
        c[0] = keep + dpp_mov_f64<DPP_QUAD_XOR2>(send);
      }
      const int b0 = (h2 ? 4 : 0) + (h0 ? 2 : 0) + (h1 ? 1 : 0);
      cpw[q + 32 * b0] = c[0];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = swap32_add(r[8 + e], r[12 + e]);  // rows 8 + e (lanes 0-31), 12 + e (lanes 32-63)
#pragma unroll
      for (int e = 0; e < 2; ++e) r[e] = swap16_add(r[e], r[e + 2]);
      {
        const double keep = h3 ? r[1] : r[0], send = h3 ? r[0] : r[1];
        r[0] = keep + dpp_mov_f64<DPP_ROW_ROR8>(send);
      }
      rp[p + 16 * (8 + (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0))] = r[0];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const double keep = h2 ? c[6 + b] : c[4 + b], send = h2 ? c[4 + b] : c[6 + b];
        c[b] = keep + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(send);
      }
      {
        const double keep = h0 ? c[1] : c[0], send = h0 ? c[0] : c[1];
        c[0] = keep + dpp_mov_f64<DPP_QUAD_XOR1>(send);
      }
      c[0] = c[0] + dpp_mov_f64<DPP_QUAD_XOR2>(c[0]);  // (the two lanes of the pair end with the same sum)
      if (!h1) cpw[q + 32 * (4 + (h2 ? 2 : 0) + (h0 ? 1 : 0))] = c[0];
    }
    dpart = wave_allsum(dpart);
    if (l == 0) S[A_RED + (t >> 6)] = dpart;
  }
  TRD_TICK(2);
  lds_barrier();
  TRD_TICK(3);
  // (5) y_i -> w_i = tau y_i - (tau^2 / 2) (v . y) v_i
  if (t < TN) {
    const double* rp = S + A_RP;
    const double* cp = S + A_CP;
    const double* red = S + A_RED;
    const double y = ((rp[t] + rp[256 + t]) + (rp[512 + t] + rp[768 + t])) + (cp[t] + cp[256 + t]);
    const double dot = 2.0 * (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
    const double alpha = -0.5 * tau * tau * dot;
    ws[t] = t > k ? fma(tau, y, alpha * vs[t]) : 0.0;
  }
  lds_barrier();
  TRD_TICK(4);
  // (6) A <- A - v w^T - w v^T   (diagonal blocks are held at half weight); one grid-row half (u) at a time
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    double vi[8], wi[8];
#pragma unroll
    for (int a = K; a < 8; ++a) {
      vi[a] = vs[p + 16 * u + 32 * a];
      wi[a] = ws[p + 16 * u + 32 * a];
    }
#pragma unroll
    for (int b = K; b < 8; ++b) {
      const double vj = vs[q + 32 * b], wj = ws[q + 32 * b];
      const double vjh = 0.5 * vj, wjh = 0.5 * wj;
#pragma unroll
      for (int a = b; a < 8; ++a)
        A[u][tidx(a, b)] = fma(-vi[a], (a == b) ? wjh : wj, fma(-wi[a], (a == b) ? vjh : vj, A[u][tidx(a, b)]));
    }
  }
  TRD_TICK(5);
}

// ---- workspace per matrix (doubles) --------------------------------------------------------------------------------
struct L256 {
  static constexpr int TNX = TN, TMX = TM;
  static constexpr long W_HS = 0;                        // 256 x 256 Householder vectors (row k = v_k)
  static constexpr long W_ZG = W_HS + (long)TN * TN;     // 256 x 128 eigenvectors of T, unnormalised ([i][c])
  static constexpr long W_TG = W_ZG + (long)TN * TM;     // d[256], e[256], tau[256]
  static constexpr long W_LG = W_TG + 3 * TN;            // lam[128], 1 / |z| [128], residual / |T| [128]
  static constexpr long W_MI = W_LG + 3 * TM;            // {|T|, pivmin, bad flag (int), ...}
  static constexpr long W_TM = W_MI + 16;                // 16 blocks x (16 x 16) triangular factors of the blocked reflectors
  static constexpr long W_PER = W_TM + 16 * 256;
};
constexpr long W_HS = L256::W_HS, W_ZG = L256::W_ZG, W_TG = L256::W_TG, W_LG = L256::W_LG, W_MI = L256::W_MI, W_TM = L256::W_TM,
               W_PER = L256::W_PER;

struct TrdDebug {
  long long* clk;            // TRD_STEP_PROFILE: batch x 16, cycles per part of a phase-A step in [8 .. 13]
  unsigned long long* work;  // profiling: += 1 per matrix this launch solved (not skipped, not rejected)
};

// ================= kernel A: tridiagonalisation (one workgroup = one CU per matrix) =================
// EMB = false: order 256 exactly (off = 0, ldn = 256: constant addressing, every step runs); true: an embedded order.
template <bool EMB>
__global__ __launch_bounds__(TNT, 1) void trd_a_kernel(const double* __restrict__ Gc, const int* __restrict__ rep,
                                                       double* __restrict__ ws, TrdDebug dbg, const TrdShape sh) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int bm = blockIdx.x;
  if (rep && rep[bm] != bm) return;  // duplicate of another matrix / frozen sketch: nothing to solve
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int wp = w >> 2, wq = w & 3, lp = l & 7, lq = l >> 3;  // (lq in the lane bits the v_permlane swaps reach)
  const int p = wp * 8 + lp, q = wq * 8 + lq;
  const int off = EMB ? sh.off : 0, ldn = EMB ? sh.ldn : TN;
  const double* G = Gc + (long)bm * ldn * ldn;
  double* wsm = ws + (long)bm * W_PER;
  double* Hs = wsm + W_HS;
  double* S = sm + L_S;
  double A[2][36];
  const double* Gsrc = G;
  if constexpr (EMB) {
    // the matrix is first copied into the 256-layout (entry (i, j) of the order-n matrix at (i + off, j + off), zeros
    // above / left of it), in the workspace region the Householder vectors will overwrite step by step -- every load below
    // has completed by then -- so that the register load keeps its constant addressing
    for (int e = t; e < TN * TN; e += TNT) {
      const int j = e >> 8, i = e & 255;
      Hs[e] = (j >= off && i >= off) ? G[(long)(j - off) * ldn + (i - off)] : 0.0;
    }
    __syncthreads();  // (global writes of this workgroup are visible to it behind the barrier)
    Gsrc = Hs;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        const int i = p + 16 * u + 32 * a, j = q + 32 * b;
        A[u][tidx(a, b)] = Gsrc[(long)j * TN + i] * (a == b ? 0.5 : 1.0);
      }
  for (int i = t; i < 768; i += TNT) S[A_XS + i] = 0.0;
  if (EMB) {
    __syncthreads();  // every thread has its elements: the rows of Hs may be rewritten
    if (t < off) {  // the skipped steps: identity reflectors, zero rows of T
      sm[L_D + t] = 0.0;
      sm[L_E + t] = 0.0;
      sm[L_TAU + t] = 0.0;
    }
    // the block of 16 reflectors that straddles `off` is read whole by the back-transformation: its skipped rows are zeros
    for (int i = t; i < (off & 15) * TN; i += TNT) Hs[(long)(off & ~15) * TN + i] = 0.0;
  }
  __syncthreads();
  long long prof[6] = {0, 0, 0, 0, 0, 0};
#define TRD_RUN(KV)                                                                           \
  for (int k_ = (EMB && off > 32 * (KV)) ? off : 32 * (KV); k_ < 32 * (KV) + 32 && k_ <= TN - 2; ++k_) \
    trd_step<KV>(A, k_, sm, t, p, q, wp, wq, Hs, prof);
  TRD_RUN(0) TRD_RUN(1) TRD_RUN(2) TRD_RUN(3) TRD_RUN(4) TRD_RUN(5) TRD_RUN(6) TRD_RUN(7)
#undef TRD_RUN
#ifdef TRD_STEP_PROFILE
  if (dbg.clk && t == 0)
    for (int i = 0; i < 6; ++i) dbg.clk[(long)bm * 16 + 8 + i] = prof[i];
#endif
  (void)prof;
  if (p == 15 && q == 31) sm[L_D + TN - 1] = 2.0 * A[1][tidx(7, 7)];
  if (t == 0) { sm[L_E + TN - 1] = 0.0; sm[L_TAU + TN - 1] = 0.0; }
  __syncthreads();
  if (t < TN) {
    wsm[W_TG + t] = sm[L_D + t];
    wsm[W_TG + TN + t] = sm[L_E + t];
    wsm[W_TG + 2 * TN + t] = sm[L_TAU + t];
  }
}

// ================= kernel T: triangular factors of the blocked reflectors =================
// Reflectors k0 .. k0 + 15 (k0 = 16 kb) as one block:  H_k0 ... H_k0+15 = I - V T V^T  with T upper triangular (LAPACK dlarft,
// forward / columnwise):  T_ii = tau_i,  T(0:i, i) = -tau_i T(0:i, 0:i) (V^T v_i).  One wave per block: V^T V by 64 MFMAs
// (both operands of an instruction are the same register: lane (kq, li) holds V[row 4 s + kq][reflector li]), then the
// 16 columns of T one after the other, row i on lane i.
__global__ __launch_bounds__(64) void trd_t_kernel(const int* __restrict__ rep, double* __restrict__ ws, const TrdShape sh) {
  __shared__ double Gs[16][17];
  __shared__ double Ts[16][17];
  const int bm = blockIdx.x >> 4, kb = blockIdx.x & 15;
  if (rep && rep[bm] != bm) return;
  if (16 * kb + 15 < sh.off) return;  // identity reflectors of an embedded order: kernel D skips the block as well
  double* wsm = ws + (long)bm * W_PER;
  const double* Hs = wsm + W_HS;
  const int l = threadIdx.x, kq = l >> 4, li = l & 15;
  const int kr = 16 * kb + li;
  const double* vrow = Hs + (long)(kr <= TN - 3 ? kr : 0) * TN;
  const bool okr = kr <= TN - 3;
  v4f64 Ga = {0.0, 0.0, 0.0, 0.0}, Gb = {0.0, 0.0, 0.0, 0.0};
  const int s_lo = (16 * kb) >> 2;  // rows below 16 kb hold no entry of these reflectors
  for (int s4 = s_lo; s4 < TN / 4; s4 += 2) {
    const double a0 = okr ? vrow[4 * s4 + kq] : 0.0;
    const double a1 = okr ? vrow[4 * s4 + 4 + kq] : 0.0;
    Ga = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, Ga, 0, 0, 0);
    Gb = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, Gb, 0, 0, 0);
  }
  Ga += Gb;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Gs[kq + 4 * r][li] = Ga[r];
    Ts[kq + 4 * r][li] = 0.0;
  }
  __syncthreads();
  if (l < 16) {  // row i = l of T, column after column (column c only needs the columns before it)
    for (int c = 0; c < 16; ++c) {
      const int kc = 16 * kb + c;
      const double tau = kc <= TN - 3 ? wsm[W_TG + 2 * TN + kc] : 0.0;
      double v = 0.0;
      if (l == c) v = tau;
      else if (l < c) {
        double acc = 0.0;
        for (int b2 = l; b2 < c; ++b2) acc = fma(Ts[l][b2], Gs[b2][c], acc);
        v = -tau * acc;
      }
      Ts[l][c] = v;  // (only this lane reads row l)
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) wsm[W_TM + kb * 256 + (kq + 4 * r) * 16 + li] = Ts[kq + 4 * r][li];
}

// ================= kernel D: certificate, then V = Q Z on the matrix cores, columns written as lam_j v_j =================
// The back-transformation is a contraction -- Z (256 x 128) <- (I - V T V^T) Z per block of 16 reflectors: S = V^T Z, C = T S,
// Z -= V C -- and v_mfma_f64_16x16x4_f64 fits it exactly.  Wave w owns the 16 columns 16 w .. 16 w + 15 of Z as 16 tiles of
// 16 x 16 in the C / D layout of the instruction (lane (kq, li), register r: row 16 T + kq + 4 r, column li): register r of
// a tile is at the same time the B operand of the K-slice of rows 16 T + 4 r .. 4 r + 3, so S needs no data movement, and
// the same holds for S in C = T S and for C in Z -= V C.  The A operands (V^T, T, V) come from LDS, one 8-byte read per
// lane and MFMA, from two copies of the reflector block laid out for conflict-free reads ([row][reflector] for V^T,
// [reflector][row], pitch 272, for V).  No cross-wave reduction, two barriers per block of 16 reflectors (staging).
constexpr int DM_VA = 0;                 // [256][16]   V as [row][reflector]
constexpr int DM_VB = 4096;              // [16][272]   V as [reflector][row]
constexpr int DM_TM = DM_VB + 16 * 272;  // [16][16]    T
constexpr int DM_TOTAL = DM_TM + 256;
__global__ __launch_bounds__(TNT, 1) void trd_d_kernel(double* __restrict__ Gc, const int* __restrict__ rep,
                                                       int* __restrict__ done, double* __restrict__ ws, TrdDebug dbg,
                                                       const TrdShape sh, double* __restrict__ lam_out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int bm = blockIdx.x;
  if (rep && rep[bm] != bm) {  // (the Jacobi skips it too)
    if (threadIdx.x == 0) done[bm] = 1;
    return;
  }
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int kq = l >> 4, li = l & 15;
  const int off = sh.off, ldn = sh.ldn, n = sh.n;
  double* G = Gc + (long)bm * ldn * ldn;
  double* wsm = ws + (long)bm * W_PER;
  const double* Hs = wsm + W_HS;
  const double* Zg = wsm + W_ZG;
  double* S = sm + L_S;
  int* badflag = reinterpret_cast<int*>(sm + L_MISC + 8);
  if (t < TM) {
    sm[L_LAM + t] = wsm[W_LG + t];
    sm[L_ZS + t] = wsm[W_LG + TM + t];
    sm[L_D + t] = wsm[W_LG + 2 * TM + t];  // residuals (kernel C)
  }
  if (t == 0) *badflag = reinterpret_cast<const int*>(wsm + W_MI + 2)[0];
  __syncthreads();
  const double lam0 = sm[L_LAM], lamcut = sm[L_LAM + sh.need - 1];
  auto significant = [&](int c) -> bool { return c < sh.nvec && trd_significant(sm[L_LAM + c], c, lam0, lamcut, sh); };
  if (t < 64) {  // largest residual among the significant vectors (widens the cluster rule)
    double rmax = fmax(significant(t) ? sm[L_D + t] : 0.0, significant(t + 64) ? sm[L_D + t + 64] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, o));
    if (t == 0) sm[L_MISC + 10] = rmax;
  }
  __syncthreads();
  // certificate: cosines between neighbours in the spectrum, clusters wider than the neighbourhood
  {
    const int c = t >> 2, dl = (t & 3) + 1, c2 = c + dl;
    if (c2 < TM && significant(c) && significant(c2)) {
      double dotv = 0.0;
#pragma unroll 8
      for (int i = off; i < TN; ++i) dotv = fma(Zg[(long)i * TM + c], Zg[(long)i * TM + c2], dotv);
      if (!(fabs(dotv) * sm[L_ZS + c] * sm[L_ZS + c2] <= TRD_COS_MAX)) atomicOr(badflag, 2);
    }
    if ((t & 3) == 0 && c + 5 < TM && significant(c) && significant(c + 5)) {
      const double width = fmax(1e-7 * lam0, TRD_GAP_PER_RES * sm[L_MISC + 10] * wsm[W_MI]);
      if ((sm[L_LAM + c] - sm[L_LAM + c + 5]) <= width) atomicOr(badflag, 4);
    }
  }
  __syncthreads();
  if (*badflag) {  // leave G as it is: the Jacobi solver takes this matrix
    if (t == 0) done[bm] = 0;
    return;
  }
  // Z tiles of this wave: column 16 w + li, rows 16 T + kq + 4 r
  const int col = 16 * w + li;
  const bool wact = 16 * w < sh.nvec;  // (nvec is a multiple of 32: a wave's 16 columns are all formed or none is)
  v4f64 Zt[16];
  {
    const double zs = wact ? sm[L_ZS + col] : 0.0;
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) Zt[T][r] = wact ? Zg[(long)(16 * T + kq + 4 * r) * TM + col] * zs : 0.0;
  }
  double* vA = S + DM_VA;
  double* vB = S + DM_VB;
  double* tm = S + DM_TM;
  // staging of a block: 16 reflectors x 256 rows = 8 values per thread (coalesced along the row), T: 256 values
  const int kb_lo = off >> 4;  // blocks below hold identity reflectors only (embedded order)
  auto fetch = [&](int kb, double (&nx)[8], double& tx) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int idx = t + TNT * m, kr = 16 * kb + (idx >> 8);
      nx[m] = (kb >= kb_lo && kr <= TN - 3) ? Hs[(long)kr * TN + (idx & 255)] : 0.0;
    }
    tx = (kb >= kb_lo && t < 256) ? wsm[W_TM + kb * 256 + t] : 0.0;
  };
  auto store = [&](const double (&nx)[8], double tx) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int idx = t + TNT * m, rr = idx >> 8, row = idx & 255;
      vA[row * 16 + rr] = nx[m];
      vB[rr * 272 + row] = nx[m];
    }
    if (t < 256) tm[t] = tx;
  };
  double nx[8], tx;
  const int kb_top = (TN - 3) >> 4;
  fetch(kb_top, nx, tx);
  store(nx, tx);
  __syncthreads();
  for (int kb = kb_top; kb >= kb_lo; --kb) {
    fetch(kb - 1, nx, tx);
    const int Tlo = wact ? (16 * kb + 1) >> 4 : 16;  // row tiles below hold no entry of these reflectors
    // S = V^T Z_w : M = reflector, K = row, N = column
    v4f64 Sa = {0.0, 0.0, 0.0, 0.0}, Sb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int T = 0; T < 16; ++T) {
      if (T >= Tlo) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double a = vA[(16 * T + 4 * r + kq) * 16 + li];  // A[m = li][k = kq] = V[row][reflector li]
          if (T & 1) Sb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Zt[T][r], Sb, 0, 0, 0);
          else Sa = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Zt[T][r], Sa, 0, 0, 0);
        }
      }
    }
    v4f64 Sw = Sa + Sb;
    // C = T S : M = reflector i, K = reflector j, N = column; then negated for the update
    v4f64 Cw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) Cw = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[li * 16 + 4 * s4 + kq], Sw[s4], Cw, 0, 0, 0);
    Cw = -Cw;
    // Z_tile -= V_tile C : M = row, K = reflector, N = column
#pragma unroll
    for (int T = 0; T < 16; ++T) {
      if (T >= Tlo) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          Zt[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(vB[(4 * s4 + kq) * 272 + 16 * T + li], Cw[s4], Zt[T], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with this block's LDS copies
    store(nx, tx);
    __syncthreads();
  }
  {
    // column c of the result: rows [0, n) = lam_c v_c (the embedded rows [off, 256) moved up), everything else of the
    // ldn x ldn matrix zeros
    const int ncol = n < sh.nvec ? n : sh.nvec;
    const bool cval = col < ncol;
    const double lc = cval ? sm[L_LAM + col] : 0.0;
    const double f = lc > 0.0 ? lc : 0.0;
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * T + kq + 4 * r - off;
        if (i >= 0) {
          if (col < ldn) G[(long)col * ldn + i] = cval ? f * Zt[T][r] : 0.0;
          if (col + TM < ldn) G[(long)(col + TM) * ldn + i] = 0.0;
        }
      }
    // the column norms the caller reads next are the eigenvalues themselves (|lam_c v_c| = lam_c): written here, the norms
    // pass over the matrix is skipped for solved matrices
    // (an eigenvalue whose square underflows is written as zero, as the norm of its column would come out: the spectrum of
    // an all-zero matrix is then exactly zero)
    if (lam_out && t < ldn) {
      const double lv = t < ncol ? sm[L_LAM + t] : 0.0;
      lam_out[(long)bm * ldn + t] = (lv > 0.0 && lv * lv > 0.0) ? lv : 0.0;
    }
    const int npad = ldn - n;  // padding rows of the Jacobi's layout
    for (int idx = t; idx < npad * ldn; idx += TNT) {
      const int c = idx / npad;
      G[(long)c * ldn + n + (idx - c * npad)] = 0.0;
    }
  }
  if (t == 0) {
    done[bm] = 1;
    if (dbg.work) atomicAdd(dbg.work, 1ull);
  }
}

}  // namespace mused

namespace mused {

size_t trd_workspace_doubles(int batch) { return (size_t)batch * (size_t)W_PER; }

constexpr int L_A_TOTAL = L_S + 2314;            // kernel A: persistent part + its scratch
constexpr int L_DM_TOTAL = L_S + DM_TOTAL;       // kernel D
constexpr int C_LDS = trd_c_lds_doubles<L256, 32>();        // kernel C: T, exchange, the pivot sequences of 32 vectors

int trd_prepare() {
  static std::once_flag once;
  static hipError_t rc = hipSuccess;
  std::call_once(once, [] {
    rc = hipFuncSetAttribute((const void*)trd_d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * L_DM_TOTAL));
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute((const void*)trd_c_kernel<L256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * C_LDS));
  });
  MUSED_CHECK_HIP(rc);
  return MUSED_OK;
}

// The tridiagonalisation alone on `batch` symmetric matrices of order 256 (G: batch x 256 x 256 column-major; only the lower block
// triangle of 32 x 32 blocks is read, the diagonal blocks whole).  The blocked solver (trdx.hip) hands the trailing 256 x 256 of its
// larger orders over to it.  ws: batch x trd_tail_ws_per() doubles: Householder vectors at trd_tail_off_hs() (row k = v_k, 256 x
// 256; row 255 is not written), d / e / tau (256 each) at trd_tail_off_tg().
long trd_tail_ws_per() { return W_PER; }
long trd_tail_off_hs() { return W_HS; }
long trd_tail_off_tg() { return W_TG; }
int trd_tail_launch(const double* G, const int* rep, double* ws, int batch, hipStream_t st) {
  TrdDebug dbg{nullptr, nullptr};
  TrdShape sh;
  sh.n = TN; sh.ldn = TN; sh.off = 0; sh.nvec = TM; sh.need = TM; sh.cert_all = 0;
  hipLaunchKernelGGL(trd_a_kernel<false>, dim3(batch), dim3(TNT), sizeof(double) * L_A_TOTAL, st, G, rep, ws, dbg, sh);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

bool trd_supports(int n, int ldn, int need) { return n >= 2 && n <= TN && ldn >= n && ldn <= TN && need >= 1 && need <= TM && need <= n; }

// Solves the matrices of Gc (batch x ldn x ldn column-major, symmetric of order n <= ldn <= 256, zero padded) in place:
// done[b] = 1 -> columns 0 .. min(n, nvec) - 1 of matrix b hold lam_j v_j for its largest eigenvalues (descending; nvec = `need`
// rounded up to a multiple of 32), every other entry zeros; done[b] = 0 -> untouched (certificate failed: solve it with the
// Jacobi).  `need`: how many leading pairs the caller reads; cert_all: see TrdShape.  ws: trd_workspace_doubles(batch)
// doubles.  Five launches on `st`.
int trd_solve(double* Gc, int n, int ldn, int need, bool cert_all, int batch, const int* rep, int* done, double* ws,
              hipStream_t st, long long* dbg_clk, unsigned long long* work, hipEvent_t after_a, double* lam_out) {
  MUSED_REQUIRE(trd_supports(n, ldn, need), "trd_solve: unsupported shape (n=%d, ld=%d, need=%d)", n, ldn, need);
  TrdDebug dbg{dbg_clk, work};
  TrdShape sh;
  sh.n = n; sh.ldn = ldn; sh.off = TN - n;
  sh.nvec = ((need + 31) / 32) * 32;
  sh.need = need;
  sh.cert_all = cert_all ? 1 : 0;
  const int nch = sh.nvec / 32;
  if (n == TN && ldn == TN) hipLaunchKernelGGL(trd_a_kernel<false>, dim3(batch), dim3(TNT), sizeof(double) * L_A_TOTAL, st, Gc, rep, ws, dbg, sh);
  else hipLaunchKernelGGL(trd_a_kernel<true>, dim3(batch), dim3(TNT), sizeof(double) * L_A_TOTAL, st, Gc, rep, ws, dbg, sh);
  if (after_a) MUSED_CHECK_HIP(hipEventRecord(after_a, st));  // profiling: the tridiagonalisation alone
  if (batch <= 64) hipLaunchKernelGGL((trd_b_kernel<128, L256>), dim3(nch * batch), dim3(128), 0, st, rep, ws, sh);
  else hipLaunchKernelGGL((trd_b_kernel<512, L256>), dim3(batch), dim3(512), 0, st, rep, ws, sh);
  hipLaunchKernelGGL((trd_c_kernel<L256, 32>), dim3(nch * batch), dim3(128), sizeof(double) * C_LDS, st, rep, ws, sh);
  hipLaunchKernelGGL(trd_t_kernel, dim3(16 * batch), dim3(64), 0, st, rep, ws, sh);
  hipLaunchKernelGGL(trd_d_kernel, dim3(batch), dim3(TNT), sizeof(double) * L_DM_TOTAL, st, Gc, rep, done, ws, dbg, sh, lam_out);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// test / diagnostic: copy the tridiagonal matrix, the eigenvalues and the residuals of matrix b out of the workspace
__global__ void trd_export_kernel(const double* __restrict__ ws, double* __restrict__ d, double* __restrict__ e,
                                  double* __restrict__ lam, double* __restrict__ res) {
  const int bm = blockIdx.x, t = threadIdx.x;
  const double* wsm = ws + (long)bm * W_PER;
  if (d) d[(long)bm * TN + t] = wsm[W_TG + t];
  if (e) e[(long)bm * TN + t] = wsm[W_TG + TN + t];
  if (t < TM) {
    if (lam) lam[(long)bm * TM + t] = wsm[W_LG + t];
    if (res) res[(long)bm * TM + t] = wsm[W_LG + 2 * TM + t];
  }
}

}  // namespace mused

using namespace mused;

// Diagnostic / unit-test entry (not part of the declared ABI): runs the direct solver alone on `batch` symmetric n x n
// matrices (device, column-major with leading dimension n, overwritten as trd_solve does) and returns the intermediate
// quantities of every phase in the 256-layout (order n embedded at the bottom right: the first 256 - n entries of d / e are
// zeros).  out_d / out_e: batch x 256 (tridiagonal), out_lam: batch x 128 (descending; the first `need` rounded up to 32 are
// formed), out_res: batch x 128, out_done: batch ints.
extern "C" int mused_debug_trd_n(double* G, int n, int need, int cert_all, int batch, double* out_d, double* out_e,
                                 double* out_lam, double* out_res, int* out_done, void* stream) {
  MUSED_REQUIRE(G && batch >= 1 && out_done, "mused_debug_trd: bad arguments");
  int rc = trd_prepare();
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  double* ws = nullptr;
  MUSED_CHECK_HIP(hipMalloc((void**)&ws, sizeof(double) * trd_workspace_doubles(batch)));
  rc = trd_solve(G, n, n, need, cert_all != 0, batch, nullptr, out_done, ws, st);
  if (!rc) hipLaunchKernelGGL(trd_export_kernel, dim3(batch), dim3(TN), 0, st, ws, out_d, out_e, out_lam, out_res);
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(ws);
  if (rc) return rc;
  MUSED_CHECK_HIP(e);
  return MUSED_OK;
}

extern "C" int mused_debug_trd(double* G, int batch, double* out_d, double* out_e, double* out_lam, double* out_res,
                               int* out_done, void* stream) {
  return mused_debug_trd_n(G, TN, TM, 0, batch, out_d, out_e, out_lam, out_res, out_done, stream);
}

// Diagnostic: average time (ms, HIP events) of `reps` direct solves of the same `batch` matrices of order n (G is restored
// from a copy before every solve; the copy is outside the timed region).
extern "C" int mused_debug_trd_time_n(const double* G, int n, int need, int batch, int reps, double* out_ms, int* out_done,
                                      long long* out_clk, void* stream) {
  MUSED_REQUIRE(G && batch >= 1 && reps >= 1 && out_ms, "mused_debug_trd_time: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc = trd_prepare();
  if (rc) return rc;
  double *ws = nullptr, *work = nullptr;
  int* done = nullptr;
  const size_t bytes = sizeof(double) * (size_t)batch * n * n;
  MUSED_CHECK_HIP(hipMalloc((void**)&ws, sizeof(double) * trd_workspace_doubles(batch)));
  MUSED_CHECK_HIP(hipMalloc((void**)&work, bytes));
  MUSED_CHECK_HIP(hipMalloc((void**)&done, sizeof(int) * (size_t)batch));
  hipEvent_t e0, e1;
  MUSED_CHECK_HIP(hipEventCreate(&e0));
  MUSED_CHECK_HIP(hipEventCreate(&e1));
  double total = 0.0;
  for (int i = 0; i <= reps && !rc; ++i) {  // the first solve is a warm-up
    MUSED_CHECK_HIP(hipMemcpyAsync(work, G, bytes, hipMemcpyDeviceToDevice, st));
    MUSED_CHECK_HIP(hipEventRecord(e0, st));
    rc = trd_solve(work, n, n, need, false, batch, nullptr, done, ws, st, out_clk);
    MUSED_CHECK_HIP(hipEventRecord(e1, st));
    MUSED_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MUSED_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (i > 0) total += ms;
  }
  *out_ms = total / reps;
  if (out_done) MUSED_CHECK_HIP(hipMemcpy(out_done, done, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(ws); (void)hipFree(work); (void)hipFree(done);
  return rc;
}

extern "C" int mused_debug_trd_time(const double* G, int batch, int reps, double* out_ms, int* out_done, long long* out_clk,
                                    void* stream) {
  return mused_debug_trd_time_n(G, TN, TM, batch, reps, out_ms, out_done, out_clk, stream);
}
