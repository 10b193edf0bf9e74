// Direct symmetric eigensolver for the Gram matrices of orders 320 .. 1024 (round 4): the FD rotation at l = 256 (BASELINE
// config 3: order 2 l = 512), the sketch query at l = 128 (order 3 l = 384, 4 l = 512 with rows pending) and at l = 256 (order
// 768 / 1024).  Same chain as
// trd.hip -- tridiagonalisation, the leading eigenvalues by multisection, their vectors by twisted factorisation,
// back-transformation, certificate with the one-sided Jacobi (eig.hip) as fallback -- but a matrix of this order (1 - 2 MB)
// is not register resident on one CU, and splitting it over several CUs would put a grid-wide exchange through L2 behind
// every one of its n columns.  So ONE workgroup per matrix keeps the LAPACK dsytrd structure instead:
//
//   A  blocked Householder tridiagonalisation (dlatrd panels of 16 columns, lower variant; 8 columns above order 512).  Inside a
//      panel the trailing matrix is NOT updated: column j is brought up to date from the panel's V / W (kept in registers, one
//      matrix row per thread, two above order 512), y = A v STREAMS the lower triangle of the panel-start matrix from L2 / Infinity Cache -- symmetric: every
//      128 x 32 tile is read once and serves the row part and the column part of the product -- and the rank-32 update
//      A -= V W^T + W V^T of a finished panel runs on the matrix cores (v_mfma_f64_16x16x4, operands staged in LDS).
//      Algorithmic bytes: n^3 / 6 doubles of symv reads per matrix (179 MB at n = 512): the kernel is bound by what one CU
//      pulls from the memory system, not by flops.  All reductions have a fixed order: results do not depend on the batch.
//   B, C  trd_common.h (the kernels of trd.hip, templated on the order).
//   certificate, then  V = Q Z  blocked as compact-WY blocks of 64 reflectors: the triangular factors T_b by dlarft from block
//      Grams formed on the matrix cores (one workgroup per block), then one workgroup per 32 columns of Z carries its slab
//      through all blocks in MFMA accumulator registers (S = V_b Z, C = T_b S, Z -= V_b^T C).
//
// The matrix is solved on a COPY (a rejected matrix goes to the Jacobi untouched), on its padded order ldn (zero rows /
// columns add zero eigenvalues below the spectrum of a PSD matrix).
#include <stdlib.h>

#include "trd_common.h"

namespace mused {

constexpr int XRB = 64;  // reflectors per compact-WY block of the back-transformation
constexpr int XTAIL = 256;  // order of the trailing matrix handed to the register-resident tridiagonalisation (trd.hip)

template <int NX>
struct LX {
  static constexpr int TNX = NX, TMX = NX / 2, NBLK = NX / XRB;
  static constexpr long W_A = 0;                                  // NX x NX   working copy of the matrix (column-major)
  static constexpr long W_HS = W_A + (long)NX * NX;               // NX x NX   Householder vectors (row k = v_k)
  static constexpr long W_ZG = W_HS + (long)NX * NX;              // NX x TMX  eigenvectors of T, unnormalised ([i][c])
  static constexpr long W_ZB = W_ZG + (long)NX * TMX;             // NX x TMX  back-transformed ([i][c])
  static constexpr long W_TG = W_ZB + (long)NX * TMX;             // d, e, tau
  static constexpr long W_LG = W_TG + 3 * NX;                     // lam, 1 / |z|, residual / |T|
  static constexpr long W_MI = W_LG + 3 * TMX;                    // {|T|, pivmin, bad flag, ...}
  static constexpr long W_TF = W_MI + 16;                         // NBLK x 64 x 64 triangular factors
  static constexpr long W_PER = W_TF + (long)NBLK * XRB * XRB;
  static_assert(W_PER % 2 == 0 && NX % 64 == 0, "layout");
};

// 32 values per lane summed over the 64 lanes of a wave: on return r[0] of lane l is the wave total of value
// idx = 16 (l >> 5 & 1) + 8 (l >> 4 & 1) + 4 (l >> 3 & 1) + 2 (l >> 2 & 1) + (l & 1); lanes that differ in bit 1 agree.
__device__ __forceinline__ double wave_treduce32(double (&r)[32], const int l, int& idx) {
#pragma unroll
  for (int e = 0; e < 16; ++e) r[e] = swap32_add(r[e], r[e + 16]);
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = swap16_add(r[e], r[e + 8]);
  const bool h3 = (l & 8) != 0, h2 = (l & 4) != 0, h0 = (l & 1) != 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double keep = h3 ? r[e + 4] : r[e], send = h3 ? r[e] : r[e + 4];
    r[e] = keep + dpp_mov_f64<DPP_ROW_ROR8>(send);
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {  // partner 7 - (l & 7) of the 8-lane group: decided by bit 2
    const double keep = h2 ? r[e + 2] : r[e], send = h2 ? r[e] : r[e + 2];
    r[e] = keep + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(send);
  }
  {
    const double keep = h0 ? r[1] : r[0], send = h0 ? r[0] : r[1];
    r[0] = keep + dpp_mov_f64<DPP_QUAD_XOR1>(send);
  }
  r[0] = r[0] + dpp_mov_f64<DPP_QUAD_XOR2>(r[0]);
  idx = ((l >> 5) & 1) * 16 + ((l >> 4) & 1) * 8 + ((l >> 3) & 1) * 4 + ((l >> 2) & 1) * 2 + (l & 1);
  return r[0];
}

// 16 values per lane: on return r[0] of lane l is the wave total of value idx = 8 b5 + 4 b4 + 2 b3 + b2 (b_i = bit i of l);
// lanes that differ in bits 0 and 1 agree.
__device__ __forceinline__ double wave_treduce16(double (&r)[16], const int l, int& idx) {
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = swap32_add(r[e], r[e + 8]);
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = swap16_add(r[e], r[e + 4]);
  const bool h3 = (l & 8) != 0, h2 = (l & 4) != 0;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const double keep = h3 ? r[e + 2] : r[e], send = h3 ? r[e] : r[e + 2];
    r[e] = keep + dpp_mov_f64<DPP_ROW_ROR8>(send);
  }
  {  // partner 7 - (l & 7) of the 8-lane group: decided by bit 2
    const double keep = h2 ? r[1] : r[0], send = h2 ? r[0] : r[1];
    r[0] = keep + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(send);
  }
  r[0] = r[0] + dpp_mov_f64<DPP_QUAD_XOR1>(r[0]);
  r[0] = r[0] + dpp_mov_f64<DPP_QUAD_XOR2>(r[0]);
  idx = ((l >> 5) & 1) * 8 + ((l >> 4) & 1) * 4 + ((l >> 3) & 1) * 2 + ((l >> 2) & 1);
  return r[0];
}

// ================= kernel A: blocked tridiagonalisation, one workgroup per matrix =================
// Orders <= 512: NX threads, thread t <-> matrix row t, panels of 16 columns.  Orders 640 .. 1024: NX / 2 threads, thread t <->
// rows t and t + NX / 2, panels of 8 columns (either way the panel's V and W rows stay in registers, 2 x 16 doubles per thread, and
// the staged panel fits the 160 KB of LDS beside the waves' partial sums).
// LDS (doubles): vs[NX] | red[NW][34] | rowv[16] roww[16] misc[8] gsum[34] | U: ypart[NW][NX] during the column steps,
// Vs[NX][NB + 1] Ws[NX][NB + 1] during the update of a finished panel.
template <int NX>
struct XA {
  static constexpr int RPT = NX > 512 ? 2 : 1;  // matrix rows per thread
  static constexpr int NT = NX / RPT;           // threads
  static constexpr int NB = NX > 512 ? 8 : 16;  // panel width
  static constexpr int NW = NT / 64;
  static constexpr int PITCH = NB + 1;
  static constexpr int VPW = NX > 512 ? 8 : 16;  // vectors per workgroup of kernel C (their pivot sequences fill its LDS)
  static constexpr int U_DOUBLES = (NW * NX > 2 * NX * PITCH) ? NW * NX : 2 * NX * PITCH;
  static constexpr int LDS_DOUBLES = NX + NW * 34 + 40 + 34 + U_DOUBLES;
  static constexpr int JEND = NX - XTAIL;       // columns reduced here; the trailing XTAIL x XTAIL goes to trd.hip
  static_assert(NT % 64 == 0 && NT <= 512 && LDS_DOUBLES * 8 <= 160 * 1024 && JEND >= NB && JEND % NB == 0, "trdx: unsupported order");
};

template <int NX>
constexpr int trdx_a_lds_doubles() {
  return XA<NX>::LDS_DOUBLES;
}

template <int NX>
__global__ __launch_bounds__(XA<NX>::NT) void trdx_a_kernel(const double* __restrict__ Gc, const int* __restrict__ rep,
                                                            double* __restrict__ ws, long long* __restrict__ prof,
                                                            double* __restrict__ g22) {
  using LY = LX<NX>;
  using K = XA<NX>;
  constexpr int RPT = K::RPT, NT = K::NT, NB = K::NB, NW = K::NW, PITCH = K::PITCH, JEND = K::JEND;
  constexpr int NS = NX / 16, NT128 = (NX + 127) / 128;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int bm = blockIdx.x;
  if (rep && rep[bm] != bm) return;
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  double* wsm = ws + (long)bm * LY::W_PER;
  double* A = wsm + LY::W_A;
  double* Hs = wsm + LY::W_HS;
  double* vs = sm;
  double* red = vs + NX;
  double* rowv = red + NW * 34;
  double* roww = rowv + 16;
  double* misc = roww + 16;
  double* gsum = misc + 8;  // [34]
  double* U = gsum + 34;
  double* ypart = U + w * NX;  // this wave's partial sums of y = A v
  {  // working copy (a rejected matrix goes to the Jacobi untouched)
    const double* G = Gc + (long)bm * NX * NX;
    for (int e = t; e < NX * NX / 2; e += NT)
      reinterpret_cast<double2*>(A)[e] = reinterpret_cast<const double2*>(G)[e];
  }
  __syncthreads();
  double Vr[RPT][NB], Wr[RPT][NB];  // rows t + h NT of the panel's V and W
#pragma unroll
  for (int h = 0; h < RPT; ++h)
#pragma unroll
    for (int c = 0; c < NB; ++c) { Vr[h][c] = 0.0; Wr[h][c] = 0.0; }
  const int kq = l >> 4, li = l & 15;
  // diagnostic (prof != nullptr): s_memtime ticks of thread 0 in [column + Householder | symv | reductions + w | panel update]
  long long pacc[4] = {0, 0, 0, 0}, plast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto ptick = [&](int slot) {
    if (prof) {
      const long long now = (long long)__builtin_amdgcn_s_memtime();
      pacc[slot] += now - plast;
      plast = now;
    }
  };
  for (int j0 = 0; j0 < JEND; j0 += NB) {
    double anext[RPT];  // column j0 of the matrix as the last panel update left it
#pragma unroll
    for (int h = 0; h < RPT; ++h) anext[h] = A[(long)j0 * NX + t + h * NT];
    for (int i = 0; i < NB; ++i) {
      const int j = j0 + i;
      // (S1) row j of the panel's V, W -> LDS; column j of the panel-start matrix (fetched a column ahead: the matrix does
      //      not change inside a panel, and the load's latency would otherwise sit in front of every column)
      double a[RPT];
#pragma unroll
      for (int h = 0; h < RPT; ++h) {
        const int row = t + h * NT;
        if (row == j) {
#pragma unroll
          for (int c = 0; c < NB; ++c) { rowv[c] = Vr[h][c]; roww[c] = Wr[h][c]; }
        }
        a[h] = (row >= j) ? anext[h] : 0.0;
      }
      if (i + 1 < NB) {
#pragma unroll
        for (int h = 0; h < RPT; ++h) anext[h] = A[(long)(j + 1) * NX + t + h * NT];
      }
      lds_barrier();
      // (S2) bring it up to date: a -= V W[j]^T + W V[j]^T over the panel's earlier columns
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c < i) {
          const double rv = rowv[c], rw = roww[c];
#pragma unroll
          for (int h = 0; h < RPT; ++h) a[h] = fma(-Vr[h][c], rw, fma(-Wr[h][c], rv, a[h]));
        }
      {
        double sq = 0.0;
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
          const int row = t + h * NT;
          if (row == j) wsm[LY::W_TG + j] = a[h];  // d_j
          if (row == j + 1) misc[0] = a[h];
          sq += (row > j + 1) ? a[h] * a[h] : 0.0;
        }
        sq = wave_allsum(sq);
        if (l == 0) red[w] = sq;
      }
      lds_barrier();
      // (S3) Householder vector (dlarfg), every thread for itself: beta = -sign(x0) |x|, tau = (beta - x0) / beta,
      //      v = x / (x0 - beta).  (A sum of squares in the denormal range is a zero column: see trd.hip.)
      double sqs = 0.0;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) sqs += red[ww];
      const double x0 = (j + 1 < NX) ? misc[0] : 0.0;
      double tau = 0.0, beta = x0, scale = 0.0;
      if (sqs > 1e-280) {
        const double hh = fma(x0, x0, sqs);
        double rs = __builtin_amdgcn_rsq(hh);
        rs = rs * fma(-0.5 * hh, rs * rs, 1.5);
        rs = rs * fma(-0.5 * hh, rs * rs, 1.5);
        const double nrm = hh * rs;
        beta = x0 >= 0.0 ? -nrm : nrm;
        tau = (beta - x0) * trd_rcp(beta);
        scale = trd_rcp(x0 - beta);
      }
      double v[RPT];
#pragma unroll
      for (int h = 0; h < RPT; ++h) {
        const int row = t + h * NT;
        v[h] = (tau != 0.0) ? (row > j + 1 ? a[h] * scale : (row == j + 1 ? 1.0 : 0.0)) : 0.0;
        vs[row] = v[h];
        Hs[(long)j * NX + row] = v[h];
#pragma unroll
        for (int c = 0; c < NB; ++c) Vr[h][c] = (c == i) ? v[h] : Vr[h][c];
      }
      if (t == 0) {
        wsm[LY::W_TG + NX + j] = (j + 1 < NX) ? beta : 0.0;  // e_j
        wsm[LY::W_TG + 2 * NX + j] = tau;
      }
#pragma unroll
      for (int e = 0; e < NX / 64; ++e) ypart[l + 64 * e] = 0.0;
      lds_barrier();
      ptick(0);
      if (tau == 0.0) {  // H = I (uniform: every thread computed the same scalars from the same data)
#pragma unroll
        for (int h = 0; h < RPT; ++h)
#pragma unroll
          for (int c = 0; c < NB; ++c) Wr[h][c] = (c == i) ? 0.0 : Wr[h][c];
        continue;
      }
      // (S4) y = A v over the lower triangle of the panel-start matrix, rows / columns > j (v is zero above).  Column
      //      strips of 16, dealt to the waves in snake order (strip lengths fall linearly); per strip row tiles of 128
      //      (two rows per lane, 16-byte loads: 16 KB in flight per wave); a tile adds  A_tile v_cols  to the rows' sums
      //      and  A_tile^T v_rows  to the strip's 16 column sums (reduced over the lanes once per strip).
      {
        const int s0 = (j + 1) >> 4, L = NS - s0;
        for (int rr = 0; rr * NW < L; ++rr) {
          const int k = rr * NW + ((rr & 1) ? NW - 1 - w : w);
          if (k >= L) continue;
          const int c0 = 16 * (s0 + k);
          double cacc[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) cacc[c] = 0.0;
          for (int Ti = c0 >> 7; Ti < NT128; ++Ti) {
            const int r0 = 128 * Ti + 2 * l;  // this lane's rows r0, r0 + 1
            const bool rin = r0 < NX;
            const double2 vrow = rin ? *reinterpret_cast<const double2*>(vs + r0) : make_double2(0.0, 0.0);
            const bool diag = 128 * Ti < c0 + 16;  // the tile reaches into the upper triangle (not maintained): mask
            double ra0 = 0.0, ra1 = 0.0;
            const double* src = A + (long)c0 * NX + (rin ? r0 : 0);
            double2 x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)
              x[c] = rin ? *reinterpret_cast<const double2*>(src + (long)c * NX) : make_double2(0.0, 0.0);
            if (!diag) {
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const double vc = vs[c0 + c];
                ra0 = fma(x[c].x, vc, ra0);
                ra1 = fma(x[c].y, vc, ra1);
                cacc[c] = fma(x[c].x, vrow.x, fma(x[c].y, vrow.y, cacc[c]));
              }
            } else {
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const int col = c0 + c;
                const double vc = vs[col];
                // lower triangle only (the upper one is not maintained); the diagonal entry belongs to the row part
                const double x0v = (r0 >= col) ? x[c].x : 0.0, x1v = (r0 + 1 >= col) ? x[c].y : 0.0;
                ra0 = fma(x0v, vc, ra0);
                ra1 = fma(x1v, vc, ra1);
                const double c0v = (r0 > col) ? x0v : 0.0, c1v = (r0 + 1 > col) ? x1v : 0.0;
                cacc[c] = fma(c0v, vrow.x, fma(c1v, vrow.y, cacc[c]));
              }
            }
            if (rin) {
              double2* yp = reinterpret_cast<double2*>(ypart + r0);
              double2 y2 = *yp;
              y2.x += ra0;
              y2.y += ra1;
              *yp = y2;
            }
          }
          int idx;
          const double cs = wave_treduce16(cacc, l, idx);
          if ((l & 3) == 0) ypart[c0 + idx] += cs;
        }
      }
      lds_barrier();
      ptick(1);
      // (S5) y0 = sum of the waves' parts;  G1 = W^T v,  G2 = V^T v,  S = v . y0  reduced over the workgroup: per wave by a
      //      transposing reduction, over the waves by the first 33 threads (every thread summing the 8 x 33 partials itself
      //      cost more LDS cycles than the barrier this takes)
      double y0[RPT];
#pragma unroll
      for (int h = 0; h < RPT; ++h) {
        y0[h] = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) y0[h] += U[ww * NX + t + h * NT];
      }
      {
        double g[32];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          double gw = 0.0, gv = 0.0;
          if (c < NB) {
#pragma unroll
            for (int h = 0; h < RPT; ++h) {
              gw = fma(Wr[h][c < NB ? c : 0], v[h], gw);
              gv = fma(Vr[h][c < NB ? c : 0], v[h], gv);
            }
          }
          g[c] = gw;
          g[16 + c] = gv;
        }
        int idx;
        const double gs = wave_treduce32(g, l, idx);
        if ((l & 2) == 0) red[w * 34 + idx] = gs;
        double sv = 0.0;
#pragma unroll
        for (int h = 0; h < RPT; ++h) sv = fma(v[h], y0[h], sv);
        const double ss = wave_allsum(sv);
        if (l == 0) red[w * 34 + 32] = ss;
      }
      lds_barrier();
      if (t < 33) {
        double sacc = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) sacc += red[ww * 34 + t];
        gsum[t] = sacc;
      }
      lds_barrier();
      // (S6) y = y0 - V G1 - W G2;  w = tau y - (tau^2 / 2) (v . y) v   with  v . y = S - 2 G1 . G2
      {
        double y[RPT], dot = 0.0;
#pragma unroll
        for (int h = 0; h < RPT; ++h) y[h] = y0[h];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          if (c < i) {
            const double g1 = gsum[c], g2 = gsum[16 + c];
#pragma unroll
            for (int h = 0; h < RPT; ++h) y[h] = fma(-Vr[h][c], g1, fma(-Wr[h][c], g2, y[h]));
            dot = fma(g1, g2, dot);
          }
        }
        const double vy = gsum[32] - 2.0 * dot;
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
          const double wv = (t + h * NT > j) ? fma(tau, y[h], -0.5 * tau * tau * vy * v[h]) : 0.0;
#pragma unroll
          for (int c = 0; c < NB; ++c) Wr[h][c] = (c == i) ? wv : Wr[h][c];
        }
      }
      ptick(2);
    }
    // ---- the finished panel: A[r0:, r0:] -= V W^T + W V^T on the lower triangle, 16 x 16 tiles on the matrix cores (the tile
    //      grid starts at the multiple of 16 at or below r0: with panels of 8 every other update also touches 8 finished rows /
    //      columns, whose entries are only ever multiplied by the zeros of later Householder vectors) ----
    const int r0p = j0 + NB;
    lds_barrier();  // (every wave has read ypart / red of the last column)
    double* Vs = U;
    double* Ws = U + NX * PITCH;
#pragma unroll
    for (int h = 0; h < RPT; ++h)
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        Vs[(t + h * NT) * PITCH + c] = Vr[h][c];
        Ws[(t + h * NT) * PITCH + c] = Wr[h][c];
      }
    lds_barrier();
    if (r0p < NX) {
      // tiles q = 0 .. Tt (Tt + 1) / 2 - 1 of the lower triangle (row-wise: q = I (I + 1) / 2 + J), tile q to wave q mod NW; a wave
      // takes four of its tiles at a time -- 16 loads in flight instead of 4: one tile at a time left this step latency bound
      const int R0 = r0p & ~15, Tt = (NX - R0) / 16, total = Tt * (Tt + 1) / 2;
      constexpr int TG = 4;
      for (int q0 = w; q0 < total; q0 += NW * TG) {
        double* tp[TG];
        int Rr[TG], Cr[TG];
        v4f64 acc[TG];
#pragma unroll
        for (int g = 0; g < TG; ++g) {
          const int q = (q0 + NW * g < total) ? q0 + NW * g : q0;  // (a spare slot repeats tile q0 and is not stored)
          int I = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
          I += ((I + 1) * (I + 2) / 2 <= q) ? 1 : 0;
          I -= (I * (I + 1) / 2 > q) ? 1 : 0;
          const int J = q - I * (I + 1) / 2;
          Rr[g] = R0 + 16 * I;
          Cr[g] = R0 + 16 * J;
          // tile in the C / D layout with M = matrix column, N = matrix row: lane (kq, li), register r <-> row R + li,
          // column Cc + kq + 4 r (lanes li read / write 16 consecutive rows of a column)
          tp[g] = A + (long)(Cr[g] + kq) * NX + Rr[g] + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[g][r] = tp[g][(long)4 * r * NX];
        }
#pragma unroll
        for (int g = 0; g < TG; ++g) {
#pragma unroll
          for (int s4 = 0; s4 < NB / 4; ++s4) {
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ws[(Cr[g] + li) * PITCH + 4 * s4 + kq], Vs[(Rr[g] + li) * PITCH + 4 * s4 + kq], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Vs[(Cr[g] + li) * PITCH + 4 * s4 + kq], Ws[(Rr[g] + li) * PITCH + 4 * s4 + kq], acc[g], 0, 0, 0);
          }
        }
#pragma unroll
        for (int g = 0; g < TG; ++g)
          if (q0 + NW * g < total) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tp[g][(long)4 * r * NX] = acc[g][r];
          }
      }
    }
    __syncthreads();  // the updated matrix is visible to the whole workgroup; U may be reused
    ptick(3);
  }
  {  // the trailing XTAIL x XTAIL as the last panel update left it, for the register-resident reduction of trd.hip (which reads
     // whole diagonal blocks: the lower triangle is mirrored -- the upper one is not maintained here)
    double* T22 = g22 + (long)bm * XTAIL * XTAIL;
    for (int e = t; e < XTAIL * XTAIL; e += NT) {
      const int jj = e / XTAIL, ii = e % XTAIL;
      T22[e] = (ii >= jj) ? A[(long)(JEND + jj) * NX + JEND + ii] : A[(long)(JEND + ii) * NX + JEND + jj];
    }
  }
  if (prof && t == 0)
    for (int q = 0; q < 4; ++q) prof[(long)bm * 4 + q] = pacc[q];
}

// d / e / tau and the Householder vectors of the trailing XTAIL columns (trd.hip's workspace layout) -> this solver's layout: row
// JEND + k of Hs = [zeros | v_k].  Grid (XTAIL, batch).
template <int NX>
__global__ __launch_bounds__(256) void trdx_tail_merge_kernel(const int* __restrict__ rep, double* __restrict__ ws,
                                                              const double* __restrict__ wst, const long tper, const long ohs,
                                                              const long otg) {
  using LY = LX<NX>;
  constexpr int JEND = XA<NX>::JEND;
  const int k = blockIdx.x, bm = blockIdx.y;
  if (rep && rep[bm] != bm) return;
  double* wsm = ws + (long)bm * LY::W_PER;
  const double* tw = wst + (long)bm * tper;
  for (int row = threadIdx.x; row < NX; row += 256)  // (the last reflector is the identity: its row is not written by trd.hip)
    wsm[LY::W_HS + (long)(JEND + k) * NX + row] = (row >= JEND && k < XTAIL - 1) ? tw[ohs + (long)k * XTAIL + row - JEND] : 0.0;
  if (k == 0) {
    const int i = threadIdx.x;  // (XTAIL == 256 threads)
    wsm[LY::W_TG + JEND + i] = tw[otg + i];
    wsm[LY::W_TG + NX + JEND + i] = tw[otg + XTAIL + i];
    wsm[LY::W_TG + 2 * NX + JEND + i] = tw[otg + 2 * XTAIL + i];
  }
}

// ================= certificate + normalisation: one workgroup per matrix =================
// act[b] = b: the matrix passed (back-transformation and store run for it); -1: skipped (duplicate / frozen) or rejected.
// done[b] as trd.hip; jrep[b] = what the Jacobi's launches test (rep[b] != b -> skip): -1 for solved matrices.
template <typename LY>
__global__ __launch_bounds__(256) void trdx_cert_kernel(const int* __restrict__ rep, double* __restrict__ ws,
                                                        int* __restrict__ done, int* __restrict__ act, int* __restrict__ jrep,
                                                        int* __restrict__ nrej, unsigned long long* __restrict__ work,
                                                        const TrdShape sh) {
  constexpr int TNX = LY::TNX, TMX = LY::TMX;
  __shared__ double lam[TMX], zs[TMX], rs[TMX];
  __shared__ int bad;
  __shared__ double rmax_s;
  const int bm = blockIdx.x, t = threadIdx.x;
  if (rep && rep[bm] != bm) {  // (the Jacobi skips it too)
    if (t == 0) { done[bm] = 1; act[bm] = -1; jrep[bm] = rep[bm]; }
    return;
  }
  double* wsm = ws + (long)bm * LY::W_PER;
  const double* Zg = wsm + LY::W_ZG;
  for (int c = t; c < sh.nvec; c += 256) {
    lam[c] = wsm[LY::W_LG + c];
    zs[c] = wsm[LY::W_LG + TMX + c];
    rs[c] = wsm[LY::W_LG + 2 * TMX + c];
  }
  if (t == 0) bad = reinterpret_cast<const int*>(wsm + LY::W_MI + 2)[0];
  __syncthreads();
  const double lam0 = lam[0], lamcut = lam[sh.need - 1];
  auto significant = [&](int c) -> bool { return c < sh.nvec && trd_significant(lam[c], c, lam0, lamcut, sh); };
  if (t == 0) {
    double rmax = 0.0;  // largest residual among the significant vectors
    for (int c = 0; c < sh.nvec; ++c) rmax = (significant(c) && rs[c] > rmax) ? rs[c] : rmax;
    rmax_s = rmax;
  }
  __syncthreads();
  const double width = fmax(1e-7 * lam0, TRD_GAP_PER_RES * rmax_s * wsm[LY::W_MI]);
  // thread c: cosines of vector c with its 4 neighbours (lanes walk consecutive columns of Zg: coalesced), cluster rule
  for (int c = t; c < sh.nvec; c += 256) {
    if (!significant(c)) continue;
    double dt[4] = {0.0, 0.0, 0.0, 0.0};
    const int nn = sh.nvec - 1 - c < 4 ? sh.nvec - 1 - c : 4;
#pragma unroll 4
    for (int i = 0; i < TNX; ++i) {
      const double* zr = Zg + (long)i * TMX + c;
      const double z0 = zr[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) dt[q] = fma(z0, q < nn ? zr[q + 1] : 0.0, dt[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nn && significant(c + q + 1) && !(fabs(dt[q]) * zs[c] * zs[c + q + 1] <= TRD_COS_MAX)) atomicOr(&bad, 2);
    if (c + 5 < sh.nvec && significant(c + 5) && (lam[c] - lam[c + 5]) <= width) atomicOr(&bad, 4);
  }
  __syncthreads();
  const bool ok = bad == 0;
  if (t == 0) {
    done[bm] = ok ? 1 : 0;
    act[bm] = ok ? bm : -1;
    jrep[bm] = ok ? -1 : bm;
    if (!ok) atomicAdd(nrej, 1);
    if (ok && work) atomicAdd(work, 1ull);
  }
}

// ================= back-transformation on the matrix cores =================
// A chunk of the Householder vectors -- reflectors 64 kb .. 64 kb + 63 (rows of Hs), matrix rows 64 c .. 64 c + 63 -- goes through
// LDS as Vs[reflector][row]; thread t carries rows 2 (t & 31), + 1 of reflectors (t >> 5) + 8 q: a load instruction reads 512
// contiguous bytes per reflector, the LDS stores are conflict free.
__device__ __forceinline__ void vchunk_load(const double* __restrict__ Hs, const int nx, const int kb, const int c, const int t,
                                            double (&r)[16]) {
  const double* p = Hs + (long)(64 * kb + (t >> 5)) * nx + 64 * c + 2 * (t & 31);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double2 v = *reinterpret_cast<const double2*>(p + (long)8 * q * nx);
    r[2 * q] = v.x;
    r[2 * q + 1] = v.y;
  }
}
__device__ __forceinline__ void vchunk_store(double* __restrict__ Vs, const int pitch, const int t, const double (&r)[16]) {
#pragma unroll
  for (int q = 0; q < 8; ++q)
    *reinterpret_cast<double2*>(Vs + ((t >> 5) + 8 * q) * pitch + 2 * (t & 31)) = make_double2(r[2 * q], r[2 * q + 1]);
}

// Triangular factors of the compact-WY blocks (LAPACK dlarft, forward / columnwise): one workgroup (256 threads) per block of 64
// reflectors and matrix.
//   H_k0 ... H_k0+63 = I - V^T T V  (V = the block's rows of Hs),  T upper triangular:  T_ii = tau_i,
//   T(0:i, i) = -tau_i T(0:i, 0:i) (V v_i)
// The block Gram  G = V V^T  (lower triangle, 10 tiles of 16 x 16) accumulates on the matrix cores: chunk by chunk through LDS,
// wave w takes the chunk's rows 16 w .. 16 w + 15 (the four partial Grams are added in wave order); then one wave builds T column
// by column.  (Rounds 3-4 ran 8 batched GEMMs with a quarter-filled 128 x 128 tile each, and a 64-thread dlarft kernel.)
constexpr int TF_VP = 66, TF_GP = 66, TF_TP = 65;
constexpr int TF_LDS_DOUBLES = 64 * TF_VP + 64 * TF_GP;

template <int NX>
__global__ __launch_bounds__(256) void trdx_tfac_kernel(const int* __restrict__ act, double* __restrict__ ws) {
  using LY = LX<NX>;
  extern __shared__ __attribute__((aligned(16))) double smt[];
  double* Vs = smt;                // [64][TF_VP]; later T^T [64][TF_TP]
  double* Gs = smt + 64 * TF_VP;   // [64][TF_GP]
  const int kb = blockIdx.x, bm = blockIdx.y, t = threadIdx.x, w = t >> 6, l = t & 63, kq = l >> 4, li = l & 15;
  if (act[bm] != bm) return;
  double* wsm = ws + (long)bm * LY::W_PER;
  const double* Hs = wsm + LY::W_HS;
  v4f64 g[10];  // tile (i, j), j <= i, at i (i + 1) / 2 + j
#pragma unroll
  for (int e = 0; e < 10; ++e) g[e] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double stg[16];
  vchunk_load(Hs, NX, kb, kb, t, stg);
  for (int c = kb; c < LY::NBLK; ++c) {
    __syncthreads();
    vchunk_store(Vs, TF_VP, t, stg);
    __syncthreads();
    if (c + 1 < LY::NBLK) vchunk_load(Hs, NX, kb, c + 1, t, stg);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double f[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) f[q] = Vs[(16 * q + li) * TF_VP + 16 * w + kq + 4 * s];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          g[i * (i + 1) / 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[i], f[j], g[i * (i + 1) / 2 + j], 0, 0, 0);
    }
  }
  for (int ww = 0; ww < 4; ++ww) {  // G = the waves' parts, added in wave order
    if (w == ww) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double* pg = Gs + (16 * i + kq + 4 * r) * TF_GP + 16 * j + li;
            const double v = g[i * (i + 1) / 2 + j][r];
            *pg = (ww == 0) ? v : *pg + v;
          }
    }
    __syncthreads();
  }
  // T column by column (column c only needs the columns before it); thread l < 64 owns row l, kept as column l of Tt
  double* Tt = Vs;
  if (w == 0) {
    for (int c = 0; c < XRB; ++c) {
      const double tau = wsm[LY::W_TG + 2 * NX + XRB * kb + c];
      double v = 0.0;
      if (l == c) v = tau;
      else if (l < c) {
        double accv = 0.0;
        for (int b2 = l; b2 < c; ++b2) accv = fma(Tt[b2 * TF_TP + l], Gs[c * TF_GP + b2], accv);
        v = -tau * accv;
      }
      Tt[c * TF_TP + l] = v;  // T[l][c]   (one wave: its LDS operations complete in order)
    }
  }
  __syncthreads();
  double* Tf = wsm + LY::W_TF + (long)kb * XRB * XRB;
  for (int e = t; e < XRB * XRB; e += 256) {
    const int r = e >> 6, c = e & 63;
    Tf[e] = Tt[c * TF_TP + r];  // row-major T[r][c]
  }
}

// Z <- H_0 H_1 ... Z: the blocks in reverse order, each  Z -= V^T (T (V Z))  on rows 64 kb ..  One workgroup per 32 (16 above order
// 512) columns of Z and matrix: the columns are independent, so a workgroup carries its slab of Z through ALL blocks in the
// accumulator registers of the matrix cores (row tile rt = 4 t + w of wave w: NX / 64 tiles of 16 rows per wave) -- the D layout
// of v_mfma_f64_16x16x4 is also its B layout, so  S = V Z  takes the tiles straight from the registers; V passes through LDS
// twice per block (S, then the update), T_b S once.  (Rounds 3-4: three batched GEMMs per block with mostly empty 128 x 128 tiles,
// 24 launches per solve that took as long as the tridiagonalisation.)
template <int NX>
struct XZ {
  static constexpr int CT = NX > 512 ? 1 : 2;  // column tiles of 16 per workgroup
  static constexpr int NC = 16 * CT;
  static constexpr int TPW = NX / 64;          // row tiles per wave
  // LDS pitches: a 32-lane half of a ds_read_b64 (k-lanes kq in {0, 1} or {2, 3}, 16 lanes li each) is conflict free when its
  // 32 addresses differ mod 32 doubles.  Vs[reflector][row], pitch 66: S = V Z reads reflector 16 mi + li, row .. + kq
  // (2 li + kq); the products whose k index is a reflector take k-step s, lane kq as reflector kmap(s, kq) = 8 kq + ..., so that
  // the update's Vs[kmap][.. + li] (16 kq + li) and S / C [kmap][16 ct + li] with pitch NC + 2 (16 kq + li) are conflict free too.
  static constexpr int VP = 66;
  static constexpr int SP = NC + 2;
  static constexpr int LDS_DOUBLES = 64 * VP + 2 * 64 * SP;
  __device__ static __forceinline__ int kmap(const int s, const int kq) { return 8 * kq + (s & 7) + 32 * (s >> 3); }
};

template <int NX>
__global__ __launch_bounds__(256) void trdx_back_kernel(const int* __restrict__ act, double* __restrict__ ws, const TrdShape sh) {
  using LY = LX<NX>;
  using K = XZ<NX>;
  constexpr int CT = K::CT, NC = K::NC, TPW = K::TPW, VP = K::VP, SP = K::SP, TMX = LY::TMX;
  extern __shared__ __attribute__((aligned(16))) double smz[];
  double* Vs = smz;
  double* Ss = smz + 64 * VP;
  double* Cs = Ss + 64 * SP;
  const int bm = blockIdx.y, t = threadIdx.x, w = t >> 6, l = t & 63, kq = l >> 4, li = l & 15;
  if (act[bm] != bm) return;
  double* wsm = ws + (long)bm * LY::W_PER;
  const double* Hs = wsm + LY::W_HS;
  const int c0 = NC * blockIdx.x;
  v4f64 z[TPW][CT];  // z[tt][ct][r] = Z[64 tt + 16 w + kq + 4 r][c0 + 16 ct + li], normalised on the way in
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = c0 + 16 * ct + li;
    const double zs = wsm[LY::W_LG + TMX + col];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[tt][ct][r] = wsm[LY::W_ZG + (long)(64 * tt + 16 * w + kq + 4 * r) * TMX + col] * zs;
  }
  for (int kb = LY::NBLK - 1; kb >= 0; --kb) {
    double tf[16];  // rows 16 w .. of T_b (A operand of C = T S)
    {
      const double* Tf = wsm + LY::W_TF + (long)kb * XRB * XRB + (16 * w + li) * XRB;
#pragma unroll
      for (int s = 0; s < 16; ++s) tf[s] = Tf[K::kmap(s, kq)];
    }
    v4f64 sp[4][CT];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) sp[mi][ct] = (v4f64){0.0, 0.0, 0.0, 0.0};
    // ---- S = V Z over rows 64 kb .. (chunk tt holds this wave's row tile tt) ----
    double stg[16];
  vchunk_load(Hs, NX, kb, kb, t, stg);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      if (tt >= kb) {
        __syncthreads();  // (the readers of the previous chunk are done)
        vchunk_store(Vs, VP, t, stg);
        __syncthreads();
        if (tt + 1 < TPW) vchunk_load(Hs, NX, kb, tt + 1, t, stg);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          double a[4];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) a[mi] = Vs[(16 * mi + li) * VP + 16 * w + kq + 4 * s];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              sp[mi][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], z[tt][ct][s], sp[mi][ct], 0, 0, 0);
        }
      }
    }
    for (int ww = 0; ww < 4; ++ww) {  // the waves' parts, added in wave order
      if (w == ww) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double* ps = Ss + (16 * mi + kq + 4 * r) * SP + 16 * ct + li;
              *ps = (ww == 0) ? sp[mi][ct][r] : *ps + sp[mi][ct][r];
            }
      }
      __syncthreads();
    }
    // ---- C = T_b S: wave w its 16 rows ----
    {
      v4f64 cc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cc[ct] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          cc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[s], Ss[K::kmap(s, kq) * SP + 16 * ct + li], cc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(16 * w + kq + 4 * r) * SP + 16 * ct + li] = cc[ct][r];
    }
    // ---- Z -= V^T C ----
    vchunk_load(Hs, NX, kb, kb, t, stg);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      if (tt >= kb) {
        __syncthreads();  // (C is complete; the readers of the previous chunk are done)
        vchunk_store(Vs, VP, t, stg);
        __syncthreads();
        if (tt + 1 < TPW) vchunk_load(Hs, NX, kb, tt + 1, t, stg);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int kr = K::kmap(s, kq);
          const double a = -Vs[kr * VP + 16 * w + li];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            z[tt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Cs[kr * SP + 16 * ct + li], z[tt][ct], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // (S and C may be rewritten)
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = c0 + 16 * ct + li;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wsm[LY::W_ZB + (long)(64 * tt + 16 * w + kq + 4 * r) * TMX + col] = z[tt][ct][r];
  }
}

// ================= result: column c of the matrix <- lam_c v_c (c < nvec), zeros elsewhere; 64 x 64 tiles through LDS ========
template <typename LY>
__global__ __launch_bounds__(256) void trdx_store_kernel(const int* __restrict__ act, const double* __restrict__ ws,
                                                         double* __restrict__ Gc, const TrdShape sh, double* __restrict__ lam_out) {
  constexpr int TNX = LY::TNX, TMX = LY::TMX, NT = TNX / 64;
  __shared__ double tile[64][65];
  const int bm = blockIdx.y;
  if (act[bm] != bm) return;
  const int ti = blockIdx.x / NT, tc = blockIdx.x % NT;  // rows 64 ti .., columns 64 tc ..
  const double* wsm = ws + (long)bm * LY::W_PER;
  const double* Zb = wsm + LY::W_ZB;
  double* G = Gc + (long)bm * TNX * TNX;
  if (lam_out && blockIdx.x == 0)  // the column norms the caller reads next: the eigenvalues themselves
    for (int cc = threadIdx.x; cc < TNX; cc += 256) {
      const double lv = cc < sh.nvec ? wsm[LY::W_LG + cc] : 0.0;
      lam_out[(long)bm * TNX + cc] = (lv > 0.0 && lv * lv > 0.0) ? lv : 0.0;  // (as the norm of the column would come out)
    }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = 64 * tc + tx;
  const double lc = c < sh.nvec ? wsm[LY::W_LG + c] : 0.0;
  const double f = lc > 0.0 ? lc : 0.0;
  for (int r = ty; r < 64; r += 4) tile[r][tx] = (c < sh.nvec) ? f * Zb[(long)(64 * ti + r) * TMX + c] : 0.0;
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) G[(long)(64 * tc + cc) * TNX + 64 * ti + tx] = tile[tx][cc];
}

template <typename LY>
__global__ void trdx_export_kernel(const double* __restrict__ ws, double* __restrict__ d, double* __restrict__ e,
                                   double* __restrict__ lam, double* __restrict__ res, int nvec) {
  const int bm = blockIdx.x, t = threadIdx.x;
  const double* wsm = ws + (long)bm * LY::W_PER;
  if (d) d[(long)bm * LY::TNX + t] = wsm[LY::W_TG + t];
  if (e) e[(long)bm * LY::TNX + t] = wsm[LY::W_TG + LY::TNX + t];
  if (t < nvec) {
    if (lam) lam[(long)bm * LY::TMX + t] = wsm[LY::W_LG + t];
    if (res) res[(long)bm * LY::TMX + t] = wsm[LY::W_LG + 2 * LY::TMX + t];
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
template <int NX>
static int trdx_prepare_t() {
  using LY = LX<NX>;
  static std::once_flag once;
  static hipError_t rc = hipSuccess;
  std::call_once(once, [] {
    rc = hipFuncSetAttribute((const void*)trdx_a_kernel<NX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)(sizeof(double) * trdx_a_lds_doubles<NX>()));
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute((const void*)trd_c_kernel<LY, XA<NX>::VPW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * trd_c_lds_doubles<LY, XA<NX>::VPW>()));
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute((const void*)trdx_tfac_kernel<NX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * TF_LDS_DOUBLES));
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute((const void*)trdx_back_kernel<NX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * XZ<NX>::LDS_DOUBLES));
  });
  MUSED_CHECK_HIP(rc);
  return trd_prepare();
}

template <int NX>
static int trdx_solve_t(double* Gc, const TrdShape& sh, int batch, const int* rep, int* done, int* act, int* jrep, int* nrej,
                        double* ws, hipStream_t st, unsigned long long* work, hipEvent_t after_a, long long* prof, double* lam_out) {
  using LY = LX<NX>;
  constexpr int VPW = XA<NX>::VPW;
  const int nvec = sh.nvec, nch32 = nvec / 32, nchc = nvec / VPW;
  double* g22 = ws + (size_t)batch * LY::W_PER;             // batch x XTAIL x XTAIL
  double* wst = g22 + (size_t)batch * XTAIL * XTAIL;        // batch x trd_tail_ws_per()
  hipLaunchKernelGGL(trdx_a_kernel<NX>, dim3(batch), dim3(XA<NX>::NT), sizeof(double) * trdx_a_lds_doubles<NX>(), st, Gc, rep, ws, prof,
                     g22);
  if (after_a) MUSED_CHECK_HIP(hipEventRecord(after_a, st));
  {
    const int rc_t = trd_tail_launch(g22, rep, wst, batch, st);
    if (rc_t) return rc_t;
  }
  hipLaunchKernelGGL(trdx_tail_merge_kernel<NX>, dim3(XTAIL, batch), dim3(256), 0, st, rep, ws, wst, trd_tail_ws_per(), trd_tail_off_hs(),
                     trd_tail_off_tg());
  hipLaunchKernelGGL((trd_b_kernel<128, LY>), dim3(nch32 * batch), dim3(128), 0, st, rep, ws, sh);
  constexpr size_t c_lds = sizeof(double) * trd_c_lds_doubles<LY, VPW>();
  hipLaunchKernelGGL((trd_c_kernel<LY, VPW>), dim3(nchc * batch), dim3(128), c_lds, st, rep, ws, sh);
  hipLaunchKernelGGL(trdx_cert_kernel<LY>, dim3(batch), dim3(256), 0, st, rep, ws, done, act, jrep, nrej, work, sh);
  MUSED_LAUNCH_CHECK();
  // triangular factors of the compact-WY blocks, then Z <- H_0 H_1 ... Z (one workgroup per 32 / 16 columns of Z)
  hipLaunchKernelGGL(trdx_tfac_kernel<NX>, dim3(LY::NBLK, batch), dim3(256), sizeof(double) * TF_LDS_DOUBLES, st, act, ws);
  hipLaunchKernelGGL(trdx_back_kernel<NX>, dim3(nvec / XZ<NX>::NC, batch), dim3(256), sizeof(double) * XZ<NX>::LDS_DOUBLES, st, act,
                     ws, sh);
  hipLaunchKernelGGL(trdx_store_kernel<LY>, dim3((NX / 64) * (NX / 64), batch), dim3(256), 0, st, act, ws, Gc, sh, lam_out);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

bool trdx_supports(int ldn, int need) {
  const bool order = (ldn >= 320 && ldn <= 512 && ldn % 64 == 0) || (ldn >= 640 && ldn <= 1024 && ldn % 128 == 0);
  return order && need >= 1 && ((need + 31) / 32) * 32 <= ldn / 2;
}

// EXPR(NX) for the supported order ldn (the callers have checked trdx_supports)
#define TRDX_DISPATCH(ldn, EXPR) \
  switch (ldn) {                 \
    case 320: EXPR(320); break;  \
    case 384: EXPR(384); break;  \
    case 448: EXPR(448); break;  \
    case 512: EXPR(512); break;  \
    case 640: EXPR(640); break;  \
    case 768: EXPR(768); break;  \
    case 896: EXPR(896); break;  \
    default: EXPR(1024); break;  \
  }

size_t trdx_workspace_doubles(int ldn, int batch) {
#define TRDX_WS(NX) return (size_t)batch * ((size_t)LX<NX>::W_PER + (size_t)XTAIL * XTAIL + (size_t)trd_tail_ws_per())
  TRDX_DISPATCH(ldn, TRDX_WS)
  return 0;
}

int trdx_prepare(int ldn) {
  MUSED_REQUIRE(trdx_supports(ldn, 1), "trdx_prepare: unsupported order %d", ldn);
#define TRDX_PREP(NX) return trdx_prepare_t<NX>()
  TRDX_DISPATCH(ldn, TRDX_PREP)
  return MUSED_OK;
}

// Solves the matrices of Gc (batch x ldn x ldn column-major, symmetric, zero padded beyond the caller's order) in place:
// done[b] = 1 -> columns 0 .. nvec - 1 of matrix b hold lam_j v_j for its largest eigenvalues (descending; nvec = `need` rounded up
// to a multiple of 32), every other entry zeros; done[b] = 0 -> untouched (certificate failed: the Jacobi solves it; jrep[b] = b).
// act / jrep: batch ints each (device); *nrej (device int, the caller clears it) += matrices rejected.
// ws: trdx_workspace_doubles(ldn, batch).
int trdx_solve(double* Gc, int ldn, int need, bool cert_all, int batch, const int* rep, int* done, int* act, int* jrep,
               int* nrej, double* ws, hipStream_t st, unsigned long long* work, hipEvent_t after_a, long long* prof,
               double* lam_out) {
  MUSED_REQUIRE(trdx_supports(ldn, need), "trdx_solve: unsupported shape (order %d, need %d)", ldn, need);
  TrdShape sh;
  sh.n = ldn; sh.ldn = ldn; sh.off = 0;
  sh.nvec = ((need + 31) / 32) * 32;
  sh.need = need;
  sh.cert_all = cert_all ? 1 : 0;
#define TRDX_SOLVE(NX) return trdx_solve_t<NX>(Gc, sh, batch, rep, done, act, jrep, nrej, ws, st, work, after_a, prof, lam_out)
  TRDX_DISPATCH(ldn, TRDX_SOLVE)
  return MUSED_OK;
}

}  // namespace mused

using namespace mused;

// Diagnostic / unit-test entry (not part of the declared ABI): the solver alone on `batch` symmetric matrices of order
// n in {320, ..., 512 step 64; 640, ..., 1024 step 128} (device, column-major, overwritten as trdx_solve does).  out_d / out_e: batch x n, out_lam / out_res:
// batch x n / 2 (the first `need` rounded up to 32 are formed), out_done: batch ints.
extern "C" int mused_debug_trdx(double* G, int n, int need, int cert_all, int batch, double* out_d, double* out_e,
                                double* out_lam, double* out_res, int* out_done, void* stream) {
  MUSED_REQUIRE(G && batch >= 1 && out_done && trdx_supports(n, need), "mused_debug_trdx: bad arguments");
  int rc = trdx_prepare(n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  double* ws = nullptr;
  int* ib = nullptr;
  MUSED_CHECK_HIP(hipMalloc((void**)&ws, sizeof(double) * trdx_workspace_doubles(n, batch)));
  MUSED_CHECK_HIP(hipMalloc((void**)&ib, sizeof(int) * (2 * (size_t)batch + 1)));
  MUSED_CHECK_HIP(hipMemsetAsync(ib + 2 * batch, 0, sizeof(int), st));
  rc = trdx_solve(G, n, need, cert_all != 0, batch, nullptr, out_done, ib, ib + batch, ib + 2 * batch, ws, st, nullptr, nullptr);
  if (!rc) {
    const int nvec = ((need + 31) / 32) * 32;
#define TRDX_EXPORT(NX) hipLaunchKernelGGL((trdx_export_kernel<LX<NX>>), dim3(batch), dim3(NX), 0, st, ws, out_d, out_e, out_lam, out_res, nvec)
    TRDX_DISPATCH(n, TRDX_EXPORT)
  }
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(ws);
  (void)hipFree(ib);
  if (rc) return rc;
  MUSED_CHECK_HIP(e);
  return MUSED_OK;
}

// Diagnostic: average time (ms, HIP events) of `reps` solves of the same `batch` matrices (restored from a copy before every
// solve, outside the timed region); out_a_ms: the part up to the end of the tridiagonalisation kernel.
extern "C" int mused_debug_trdx_time(const double* G, int n, int need, int batch, int reps, double* out_ms, double* out_a_ms,
                                     int* out_done, long long* out_prof, void* stream) {
  MUSED_REQUIRE(G && batch >= 1 && reps >= 1 && out_ms && trdx_supports(n, need), "mused_debug_trdx_time: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc = trdx_prepare(n);
  if (rc) return rc;
  double *ws = nullptr, *work = nullptr;
  int* ib = nullptr;
  const size_t bytes = sizeof(double) * (size_t)batch * n * n;
  MUSED_CHECK_HIP(hipMalloc((void**)&ws, sizeof(double) * trdx_workspace_doubles(n, batch)));
  MUSED_CHECK_HIP(hipMalloc((void**)&work, bytes));
  MUSED_CHECK_HIP(hipMalloc((void**)&ib, sizeof(int) * (3 * (size_t)batch + 1)));
  MUSED_CHECK_HIP(hipMemset(ib + 3 * batch, 0, sizeof(int)));
  hipEvent_t e0, e1, ea;
  MUSED_CHECK_HIP(hipEventCreate(&e0));
  MUSED_CHECK_HIP(hipEventCreate(&e1));
  MUSED_CHECK_HIP(hipEventCreate(&ea));
  double total = 0.0, total_a = 0.0;
  for (int i = 0; i <= reps && !rc; ++i) {  // the first solve is a warm-up
    MUSED_CHECK_HIP(hipMemcpyAsync(work, G, bytes, hipMemcpyDeviceToDevice, st));
    MUSED_CHECK_HIP(hipEventRecord(e0, st));
    rc = trdx_solve(work, n, need, false, batch, nullptr, ib, ib + batch, ib + 2 * batch, ib + 3 * batch, ws, st, nullptr, ea, out_prof);
    MUSED_CHECK_HIP(hipEventRecord(e1, st));
    MUSED_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f, msa = 0.f;
    MUSED_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    MUSED_CHECK_HIP(hipEventElapsedTime(&msa, e0, ea));
    if (i > 0) { total += ms; total_a += msa; }
  }
  *out_ms = total / reps;
  if (out_a_ms) *out_a_ms = total_a / reps;
  if (out_done) MUSED_CHECK_HIP(hipMemcpy(out_done, ib, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(ea);
  (void)hipFree(ws); (void)hipFree(work); (void)hipFree(ib);
  return rc;
}
