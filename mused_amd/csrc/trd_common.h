// Pieces the two direct eigensolvers share (trd.hip: orders <= 256, matrix in registers; trdx.hip: orders 320 .. 1024, blocked,
// matrix streamed): the division-free Sturm count, kernel B (eigenvalues of T by multisection), kernel C (eigenvectors of T
// by twisted factorisation) -- templated on the order through a compile-time workspace layout -- and the certificate levels.
#pragma once
#include <math.h>

#include <type_traits>

#include "internal.h"
#include "wave_ops.h"

namespace mused {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Certificate levels.  A significant vector must have a twisted-factorisation residual |(T - lam I) z| / (|z| |T|) <= TRD_RES_MAX
// (measured: ~1e-16), a cosine <= TRD_COS_MAX with each of its 4 neighbours in the spectrum, and no 6 significant eigenvalues
// may lie within max(1e-7 lam_0, TRD_GAP_PER_RES rmax |T|) (rmax = largest residual of the matrix): pairs 5 or more apart are
// then separated by more than that width, which bounds their mutual contamination 2 rmax |T| / gap by 1e-8 -- the level the
// neighbours are tested at and the parity tests assert (1e-8 sigma_1).
constexpr double TRD_RES_MAX = 1e-13, TRD_COS_MAX = 1e-8, TRD_GAP_PER_RES = 2e8;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL stores (s_waitcnt
// vmcnt(0)): inside the step loops that would put the latency of a store nobody reads before a later phase (the
// Householder vectors, the tridiagonal eigenvectors) -- or of a prefetch -- in front of every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double trd_rcp(double y) {  // reciprocal to rounding: hardware seed (~5e-8) + two Newton steps
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  return r;
}

// Shape of a solve.  trd.hip: orders n < 256 are EMBEDDED at the bottom right of the 256-layout (rows / columns [off, 256),
// off = 256 - n): the first `off` Householder steps are identities and are skipped, and the steps that run start with the
// register blocks above / left of the matrix already inactive (K = k / 32) -- an order-n solve costs what the last n columns of
// an order-256 solve cost.  T then has `off` leading zero rows (decoupled zero eigenvalues: the Gram matrices here are PSD).
// trdx.hip: n = ldn = the padded order, off = 0.
struct TrdShape {
  int n, ldn, off;  // order, leading dimension of the matrices in Gc (column-major, n <= ldn), layout order - n
  int nvec;         // eigenpairs formed: a multiple of 32
  int need;         // the caller reads the `need` largest pairs (<= nvec)
  int cert_all;     // 1: every pair below `need` with lam > 1e-24 lam_0 is certified (the eigenstep uses all of them);
                    // 0: the pairs whose energy survives the FD shrink by lam_{need-1} (discard level 1e-10 lam_0)
};

// Which of the computed pairs the certificate covers (kernels C and D / the certificate kernel agree on this).
__device__ __forceinline__ bool trd_significant(double lc, int c, double lam0, double lamcut, const TrdShape& sh) {
  if (c >= sh.need || !(lc > 0.0)) return false;
  const double l0 = lam0 > 0.0 ? lam0 : 0.0;
  return sh.cert_all ? (lc > 1e-24 * l0) : ((lc - lamcut) > 1e-10 * l0);
}

// Workspace of one matrix (doubles) as far as kernels B and C are concerned; TNX = order of T, TMX = eigenpairs at most.
//   W_ZG  TNX x TMX   eigenvectors of T, unnormalised ([i][c])
//   W_TG  d[TNX], e[TNX], tau[TNX]
//   W_LG  lam[TMX], 1 / |z| [TMX], residual / |T| [TMX]
//   W_MI  {|T|, pivmin, bad flag (int), ...}
// A layout type provides these offsets, W_PER (stride between matrices) and TNX / TMX.

// Number of eigenvalues of T below x without a division: the sign changes of the leading principal minors
//   p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2},   p_{-1} = 1, p_0 = d_0 - x
// (p_i / p_{i-1} is the pivot q_i of the LDL^T of T - x I: a sign change is a negative pivot; the dependent chain per row is
// ONE fma instead of a reciprocal, its Newton step and an fma).  dd2[i] = {d_i, e_{i-1}^2} of T SCALED by a power of two to
// |T| in [1/2, 1) with e^2 floored at 2^-120 (an absolute perturbation of 2^-60 |T| of an off-diagonal entry, far below the
// reduction's own error): then |d_i - x| <= 2 and e^2 <= 1, a minor grows by at most 3 x per row, a pair of consecutive
// minors shrinks by at most 2^-120 per two rows from the floor (both never vanish: e^2 > 0) and 2^-53 per row from
// cancellation, and rescaling the pair by the power of two of its larger member every 8 rows keeps it within 2^(+-910) of 1.  An exact zero minor counts once with its successor (whose
// sign is then -sign of its predecessor): the same count as LAPACK's "zero pivot = negative pivot".  The sign changes go
// through a shift register of sign bits (one v_alignbit per row) that is emptied by a population count every 24 rows.
// TNX = order of T (a multiple of 8; dd2 has 8 spare entries behind row TNX - 1).
template <int TNX>
__device__ __forceinline__ int trd_sturm(const double2* __restrict__ dd2, double x) {
  static_assert(TNX % 8 == 0 && TNX >= 24, "trd_sturm: the order must be a multiple of 8");
  double p0 = 1.0, p1 = dd2[0].x - x;
  unsigned bits = (unsigned)__double2hiint(p1) >> 31;  // shift register of the minors' sign bits, newest in bit 0
  int cnt = (int)bits;                                  // (p_{-1} = 1 > 0)
  auto row = [&](const double2 de) {
    const double p2 = fma(de.x - x, p1, -(de.y * p0));
    bits = __builtin_amdgcn_alignbit(bits, (unsigned)__double2hiint(p2), 31);  // (bits << 1) | sign(p2)
    p0 = p1;
    p1 = p2;
  };
  auto rescale = [&]() {  // both minors times 2^(1023 - larger biased exponent): integer field arithmetic + two multiplications
    const unsigned e1 = __builtin_amdgcn_ubfe((unsigned)__double2hiint(p1), 20, 11), e0 = __builtin_amdgcn_ubfe((unsigned)__double2hiint(p0), 20, 11);
    const double f = __hiloint2double((int)((2046u - max(e1, e0)) << 20), 0);  // (a zero minor has field 0: the other one decides)
    p1 *= f;
    p0 *= f;
  };
  auto flush = [&](unsigned mask) {  // sign changes among the rows shifted in since the last flush (bit 0 stays as "previous sign")
    cnt += __builtin_popcount((bits ^ (bits >> 1)) & mask);
    bits &= 1u;
  };
  // rows 1 .. TNX - 8 in TNX / 8 - 1 blocks of 8 (one rescale each), the next block's entries loaded before the current
  // block's chain, then rows TNX - 7 .. TNX - 1
  constexpr int NBLK = TNX / 8 - 1, NLOOP = NBLK - 1;        // 256: 31 blocks, 30 of them in the loop
  constexpr int PENDING = (NLOOP % 3) * 8 + 15;               // rows in the register at the end: <= 31
  double2 cur[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cur[j] = dd2[1 + j];
#pragma unroll 3
  for (int b = 0; b < NLOOP; ++b) {
    double2 nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nxt[j] = dd2[9 + 8 * b + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) row(cur[j]);
    rescale();
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    if (b % 3 == 2) flush(0x00ffffffu);  // 24 rows: bits 0 .. 23 against bits 1 .. 24
  }
  {
    double2 nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nxt[j] = dd2[TNX - 7 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) row(cur[j]);  // rows TNX - 15 .. TNX - 8
    rescale();
#pragma unroll
    for (int j = 0; j < 7; ++j) row(nxt[j]);  // rows TNX - 7 .. TNX - 1
  }
  flush((PENDING >= 32) ? 0xffffffffu : ((1u << PENDING) - 1u));
  return cnt;
}

// ================= kernel B: the largest eigenvalues of T, NTB / 4 per workgroup =================
// Sturm counts at 512 equispaced points first (9 bits for every eigenvalue), then 19 passes of 5-section by the 4 lanes of a
// quad (2.32 bits each).  NTB = 512: one workgroup per matrix and 128 eigenvalues, two waves per SIMD (full vector-ALU rate:
// large batches of order-256 matrices); NTB = 128: nvec / 32 workgroups per matrix (small batches spread over the CUs they
// would leave idle; each evaluates all 512 points itself).  Both variants evaluate the same points in the same arithmetic: the
// eigenvalues do not depend on the batch size (lock-step lanes == single sketches bit for bit).
template <int NTB, typename LY>
__global__ __launch_bounds__(NTB) void trd_b_kernel(const int* __restrict__ rep, double* __restrict__ ws, const TrdShape sh) {
  constexpr int TNX = LY::TNX, NWV = NTB / 64;
  const int NCH = NTB == 512 ? 1 : sh.nvec / (NTB / 4);  // workgroups per matrix (NTB / 4 eigenvalues each)
  constexpr int PASSES = 19, NPT = 512;  // 5^19 x 513 > 2^53
  __shared__ __attribute__((aligned(16))) double2 dd2[TNX + 8];
  __shared__ double part[3 * NWV];
  __shared__ double scal[4];
  __shared__ int cnts[NPT];
  const int bm = blockIdx.x / NCH, cq = blockIdx.x % NCH;
  if (rep && rep[bm] != bm) return;
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  double* wsm = ws + (long)bm * LY::W_PER;
  const double* dg = wsm + LY::W_TG;
  const double* eg = wsm + LY::W_TG + TNX;
  double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308, e2m = 0.0;
  for (int i = t; i < TNX; i += NTB) {
    const double di = dg[i], em = i > 0 ? eg[i - 1] : 0.0, ep = eg[i];
    const double rad = fabs(em) + fabs(ep);
    lo = fmin(lo, di - rad);
    hi = fmax(hi, di + rad);
    e2m = fmax(e2m, ep * ep);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
    e2m = fmax(e2m, __shfl_xor(e2m, o));
  }
  if (l == 0) { part[3 * w] = lo; part[3 * w + 1] = hi; part[3 * w + 2] = e2m; }
  __syncthreads();
  if (t == 0) {
    for (int ww = 1; ww < NWV; ++ww) { lo = fmin(lo, part[3 * ww]); hi = fmax(hi, part[3 * ww + 1]); e2m = fmax(e2m, part[3 * ww + 2]); }
    const double tn = fmax(fabs(lo), fabs(hi));
    const double piv = 2.2250738585072014e-308 * fmax(1.0, e2m);
    // power-of-two scale that brings |T| (Gershgorin) into [1/2, 1): exact, the eigenvalues are scaled back at the end
    int ks = (tn > 0.0 && tn < 1.7976931348623157e308) ? __builtin_amdgcn_frexp_exp(tn) : 0;
    ks = ks > 1000 ? 1000 : (ks < -1000 ? -1000 : ks);  // (both 2^ks and 2^-ks stay finite)
    const double sc = __builtin_amdgcn_ldexp(1.0, -ks);
    scal[0] = (lo - 2.0 * tn * 2.220446049250313e-16 * TNX - 2.0 * piv) * sc;
    scal[1] = (hi + 2.0 * tn * 2.220446049250313e-16 * TNX + 2.0 * piv) * sc;
    scal[2] = sc;
    scal[3] = __builtin_amdgcn_ldexp(1.0, ks);
    if (cq == 0) {
      wsm[LY::W_MI] = tn;
      wsm[LY::W_MI + 1] = piv;
      reinterpret_cast<int*>(wsm + LY::W_MI + 2)[0] = 0;  // bad flag (kernel C raises it)
    }
  }
  __syncthreads();
  const double gl = scal[0], gu = scal[1], sc = scal[2], unsc = scal[3];
  for (int i = t; i < TNX; i += NTB) {
    const double es_ = (i > 0 ? eg[i - 1] : 0.0) * sc;
    dd2[i] = make_double2(dg[i] * sc, fmax(es_ * es_, 7.52316384526264e-37));  // 2^-120
  }
  if (t < 8) dd2[TNX + t] = make_double2(0.0, 0.0);  // (read ahead by the row loop, never used)
  __syncthreads();
  const double h0 = (gu - gl) * (1.0 / (double)(NPT + 1));
  for (int i = t; i < NPT; i += NTB) cnts[i] = trd_sturm<TNX>(dd2, fma(h0, (double)(i + 1), gl));
  __syncthreads();
  const int r = cq * (NTB / 4) + (t >> 2), s = t & 3, jidx = TNX - 1 - r;  // r-th largest = ascending index jidx
  if (r >= sh.nvec) return;  // (whole quads, and no barrier follows)
  int first = 0;  // smallest point index whose count exceeds jidx (NPT: none) -- counts are non-decreasing
  for (int step = NPT / 2; step > 0; step >>= 1)
    if (first + step <= NPT && cnts[first + step - 1] <= jidx) first += step;
  if (first < NPT && cnts[first] <= jidx) first += 1;
  lo = first == 0 ? gl : fma(h0, (double)first, gl);
  hi = first >= NPT ? gu : fma(h0, (double)(first + 1), gl);
  for (int it = 0; it < PASSES; ++it) {
    const double h = (hi - lo) * 0.2;
    const double x = fma(h, (double)(s + 1), lo);
    const int above = trd_sturm<TNX>(dd2, x) > jidx ? 0 : 1;  // 1: the eigenvalue is >= x
    int nf = above;
    nf += __builtin_amdgcn_mov_dpp(nf, DPP_QUAD_XOR1, 0xf, 0xf, false);
    nf += __builtin_amdgcn_mov_dpp(nf, DPP_QUAD_XOR2, 0xf, 0xf, false);
    const double nlo = fma(h, (double)nf, lo);
    hi = (nf == 4) ? hi : fma(h, (double)(nf + 1), lo);
    lo = nlo;
  }
  if (s == 0) wsm[LY::W_LG + r] = 0.5 * (lo + hi) * unsc;
}

// ================= kernel C: eigenvectors of T by twisted factorisation, VPW per workgroup =================
// Two waves: wave 0 runs the forward pivots and the part of each vector above its twist index, wave 1 the backward pivots
// and the part below (two dependent chains of TNX - 1 divisions side by side); lane c < VPW of either wave = vector
// cq * VPW + c.  The pivot sequences of the VPW vectors stay in LDS ([i][vector], 2 x TNX x VPW doubles: one workgroup per
// CU; VPW = 32 at order 256, 16 up to order 512, 8 up to order 1024); every stretch of a dependent chain is preceded by its batch of loads.
template <typename LY, int VPW>
constexpr int trd_c_lds_doubles() { return 2 * LY::TNX + LY::TNX + 4 * VPW + 2 * LY::TNX * VPW; }

template <typename LY, int VPW>
__global__ __launch_bounds__(128) void trd_c_kernel(const int* __restrict__ rep, double* __restrict__ ws, const TrdShape sh) {
  constexpr int TNX = LY::TNX, TMX = LY::TMX;
  static_assert(TNX % 32 == 0 && (VPW == 8 || VPW == 16 || VPW == 32), "trd_c_kernel: unsupported shape");
  extern __shared__ __attribute__((aligned(16))) double smc[];  // trd_c_lds_doubles<LY, VPW>() doubles
  double2* dd2 = reinterpret_cast<double2*>(smc);  // [TNX]
  double* es = smc + 2 * TNX;                      // [TNX]
  double* xch = es + TNX;                          // [VPW * 4]
  double* qp = xch + 4 * VPW;                      // forward pivots [TNX][VPW]
  double* qm = qp + TNX * VPW;                     // backward pivots [TNX][VPW]
  const int nch = sh.nvec / VPW;  // workgroups per matrix
  const int bm = blockIdx.x / nch, cq = blockIdx.x - bm * nch;
  if (rep && rep[bm] != bm) return;
  const int t = threadIdx.x, role = t >> 6, l = t & 63;
  double* wsm = ws + (long)bm * LY::W_PER;
  double* Zg = wsm + LY::W_ZG;
  for (int i = t; i < TNX; i += 128) {
    const double em = i > 0 ? wsm[LY::W_TG + TNX + i - 1] : 0.0;
    dd2[i] = make_double2(wsm[LY::W_TG + i], em * em);
    es[i] = wsm[LY::W_TG + TNX + i];
  }
  __syncthreads();
  const double tnorm = wsm[LY::W_MI], pivmin = wsm[LY::W_MI + 1];
  const double lam0 = wsm[LY::W_LG], lamcut = wsm[LY::W_LG + sh.need - 1];
  const bool act = l < VPW;
  const int cl = l & (VPW - 1), c = cq * VPW + cl;  // (pivot arrays: column cl of this workgroup's VPW)
  const double lam = wsm[LY::W_LG + c];
  auto guard = [&](double v) -> double { return fabs(v) < pivmin ? -pivmin : v; };
  // the dependent chains run in stretches of 8 rows (TNX - 1 = 8 q + 7: q full stretches and one of 7), each preceded by
  // its loads
  if (act) {
    if (role == 0) {
      double qv = dd2[0].x - lam;
      qp[cl] = qv;
      auto stretch = [&](const int i0, auto cnt) {  // rows i0 .. i0 + CNT - 1
        constexpr int CNT = decltype(cnt)::value;
        double2 de[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) de[j] = dd2[i0 + j];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          qv = fma(-de[j].y, trd_rcp(guard(qv)), de[j].x - lam);
          qp[(i0 + j) * VPW + cl] = qv;
        }
      };
      for (int i0 = 1; i0 + 8 <= TNX; i0 += 8) stretch(i0, std::integral_constant<int, 8>());
      stretch(TNX - 7, std::integral_constant<int, 7>());
    } else {
      double qv = dd2[TNX - 1].x - lam;
      qm[(TNX - 1) * VPW + cl] = qv;
      auto stretch = [&](const int i0, auto cnt) {  // rows i0, i0 - 1, .. i0 - CNT + 1
        constexpr int CNT = decltype(cnt)::value;
        double dx[CNT], e2[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          dx[j] = dd2[i0 - j].x - lam;
          e2[j] = dd2[i0 - j + 1].y;
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          qv = fma(-e2[j], trd_rcp(guard(qv)), dx[j]);
          qm[(i0 - j) * VPW + cl] = qv;
        }
      };
      for (int i0 = TNX - 2; i0 >= 7; i0 -= 8) stretch(i0, std::integral_constant<int, 8>());
      stretch(6, std::integral_constant<int, 7>());
    }
  }
  __syncthreads();  // (the other wave reads the pivots)
  if (act) {  // gamma_i = qp_i + qm_i - (d_i - lam): each role scans one half, ties to the smaller index
    const int i0 = role * (TNX / 2);
    double best = 1.7976931348623157e308;
    int kt = i0;
    for (int ib = i0; ib < i0 + TNX / 2; ib += 16) {
      double a[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        a[j] = qp[(ib + j) * VPW + cl];
        b[j] = qm[(ib + j) * VPW + cl];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double g = fabs((a[j] + b[j]) - (dd2[ib + j].x - lam));
        if (g < best) { best = g; kt = ib + j; }
      }
    }
    xch[cl * 4 + role] = best;
    xch[cl * 4 + 2 + role] = (double)kt;
  }
  __syncthreads();
  int kt = 0;
  double gbest = 0.0;
  if (act) {
    const double b0 = xch[cl * 4], b1 = xch[cl * 4 + 1];
    const bool up = !(b1 < b0);  // ties: the smaller index (first half)
    gbest = up ? b0 : b1;
    kt = (int)(up ? xch[cl * 4 + 2] : xch[cl * 4 + 3]);
  }
  __syncthreads();  // xch is read: it may be rewritten
  if (act) {
    double ss = 0.0, zc = 1.0;
    if (role == 0) {
      Zg[(long)kt * TMX + c] = 1.0;
      for (int ib = kt - 1; ib >= 0; ib -= 8) {
        double qq[8], ee[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = ib - j;
          qq[j] = i >= 0 ? qp[i * VPW + cl] : 1.0;
          ee[j] = i >= 0 ? es[i] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          zc = -ee[j] * zc * trd_rcp(guard(qq[j]));
          if (ib - j >= 0) Zg[(long)(ib - j) * TMX + c] = zc;
          ss = fma(zc, zc, ss);
        }
      }
    } else {
      for (int ib = kt + 1; ib < TNX; ib += 8) {
        double qq[8], ee[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = ib + j;
          qq[j] = i < TNX ? qm[i * VPW + cl] : 1.0;
          ee[j] = i < TNX ? es[i - 1] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          zc = -ee[j] * zc * trd_rcp(guard(qq[j]));
          if (ib + j < TNX) Zg[(long)(ib + j) * TMX + c] = zc;
          ss = fma(zc, zc, ss);
        }
      }
    }
    xch[cl * 4 + role] = ss;
  }
  __syncthreads();
  if (cq == 0 && t == 0 && !(fabs(lam0) <= 1.7976931348623157e308 && tnorm <= 1.7976931348623157e308))
    atomicOr(reinterpret_cast<int*>(wsm + LY::W_MI + 2), 1);  // non-finite T or spectrum: nothing below can be trusted
  if (act && role == 0) {
    const double ss = 1.0 + xch[cl * 4] + xch[cl * 4 + 1];
    const double zs = 1.0 / sqrt(ss);
    wsm[LY::W_LG + TMX + c] = zs;
    const double res = gbest * zs / (tnorm > 0.0 ? tnorm : 1.0);
    wsm[LY::W_LG + 2 * TMX + c] = res;
    if (trd_significant(lam, c, lam0, lamcut, sh) && !(ss < 1e300 && res <= TRD_RES_MAX))
      atomicOr(reinterpret_cast<int*>(wsm + LY::W_MI + 2), 1);
  }
}

}  // namespace mused
