// fp64 MFMA GEMM core for gfx950 (v_mfma_f64_16x16x4_f64), templated on the
// storage layout of both operands, the input dtype and an epilogue functor.
//
//   C[m][n] = sum_k opA(m, k) * opB(k, n)          m < M, n < N, k in [k_lo, k_hi)
//
// Operand layouts ("K-contiguous" = the reduction index is the fast axis):
//   A_KC = true  : A stored [M][K]  (row-major, lda)      -> "N" operand
//   A_KC = false : A stored [K][M]  (lda)                  -> "T" operand
//   B_KC = true  : B stored [N][K]  (ldb)   (C = A * B^T)
//   B_KC = false : B stored [K][N]  (ldb)   (C = A * B)
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each
// wave 64 x 64 = 4 x 4 MFMA tiles, 16 accumulators of 4 f64), K-step 16 staged
// through LDS as [k][m] with a 144-double row pitch (the two k-rows a 32-lane
// ds_read_b64 group touches fall on opposite halves of the 256-B bank row), two
// LDS buffers, next tile prefetched into registers under the MFMAs.
// One f64 MFMA is 64 cycles per SIMD, 16 of them per 8 ds_read_b64: the loop is
// matrix-pipe bound by construction.
#pragma once
#include <mutex>
#include "common.h"

namespace mused {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;
constexpr int GEMM_LD = 144;  // LDS row pitch in doubles
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_BYTES = 2 /*operands*/ * 2 /*buffers*/ * GEMM_BK * GEMM_LD * 8;

struct GemmArgs {
  const void* A;
  const void* B;
  long lda, ldb;
  long strideA, strideB;  // per batch (blockIdx.z), in elements
  int M, N, K;
  int kchunk;  // split-K: batch index z covers k in [z*kchunk, min(K,(z+1)*kchunk)) when splitk != 0
  int splitk;
  int bsplit;  // batched split-K: blockIdx.z = batch entry * bsplit + split; the split covers k in [split * kchunk, ...), the epilogue
               // sees z = blockIdx.z (partial results, summed in split order by gemm_batched_splitk_reduce)
  int tiles_m, tiles_n;
  const int* rep;  // optional (batched launches): batch entry z is computed only if rep[z] == z
  int sym;  // C = A A^T (A == B, M == N) with a symmetric epilogue: only tiles on or above the diagonal are computed,
            // each off-diagonal tile is also written mirrored (epi(z, col, row, v)): half the MFMA work
};

// ---- staging -------------------------------------------------------------
// K-contiguous operand: tile rows [r0, r0+128), k in [k0, k0+16).  Thread t owns
// row (t & 127) and the 8 consecutive k's of half (t >> 7).
template <typename TIn, bool VEC>
__device__ __forceinline__ void stage_load_kc(const TIn* __restrict__ src, long ld, int r0, int nrows, int k0,
                                              int kend, TIn (&r)[8]) {
  const int m = threadIdx.x & 127;
  const int kh = threadIdx.x >> 7;
  const int row = r0 + m;
  const int k = k0 + kh * 8;
  if (row < nrows) {
    const TIn* p = src + (long)row * ld + k;
    if (VEC && k + 8 <= kend) {
      if (sizeof(TIn) == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
      } else {
        const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double2 v = q[j];
          r[2 * j] = v.x;
          r[2 * j + 1] = v.y;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (k + j < kend) ? p[j] : (TIn)0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (TIn)0;
  }
}

// (The staging registers hold the operand in its INPUT type and are converted to fp64 here, on the way into LDS: a conversion
// at load time makes every load wait for its data at once -- the fp32 similarity GEMM then exposed two global-load
// latencies per K step.)
template <typename TIn>
__device__ __forceinline__ void stage_store_kc(double* __restrict__ lds, const TIn (&r)[8]) {
  const int m = threadIdx.x & 127;
  const int kh = threadIdx.x >> 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) lds[(kh * 8 + j) * GEMM_LD + m] = (double)r[j];
}

// MN-contiguous operand stored [K][MN]: thread t owns k-row (t >> 4) and the 8
// consecutive columns starting at (t & 15) * 8.
template <typename TIn, bool VEC>
__device__ __forceinline__ void stage_load_mc(const TIn* __restrict__ src, long ld, int c0, int ncols, int k0,
                                              int kend, TIn (&r)[8]) {
  const int kk = threadIdx.x >> 4;
  const int c = c0 + (threadIdx.x & 15) * 8;
  const int k = k0 + kk;
  if (k < kend) {
    const TIn* p = src + (long)k * ld + c;
    if (VEC && c + 8 <= ncols) {
      if (sizeof(TIn) == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
      } else {
        const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double2 v = q[j];
          r[2 * j] = v.x;
          r[2 * j + 1] = v.y;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (c + j < ncols) ? p[j] : (TIn)0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (TIn)0;
  }
}

template <typename TIn>
__device__ __forceinline__ void stage_store_mc(double* __restrict__ lds, const TIn (&r)[8]) {
  const int kk = threadIdx.x >> 4;
  const int c = (threadIdx.x & 15) * 8;
  double* p = lds + kk * GEMM_LD + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = (double)r[j];
}

// Full-tile variants (all 128 rows / columns and all 16 k's inside the operand, vectorisable): no per-thread branch, so the
// loads stay in flight across the MFMAs of the current K step (with the bounds checks of the general variants the compiler
// waits for the data at the end of each conditional region).
template <typename TIn>
__device__ __forceinline__ void stage_load8(const TIn* __restrict__ p, TIn (&r)[8]) {
  if (sizeof(TIn) == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  } else {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double2 v = q[j];
      r[2 * j] = v.x;
      r[2 * j + 1] = v.y;
    }
  }
}
template <typename TIn>
__device__ __forceinline__ void stage_load_kc_full(const TIn* __restrict__ src, long ld, int r0, int k0, TIn (&r)[8]) {
  stage_load8(src + (long)(r0 + (threadIdx.x & 127)) * ld + k0 + (threadIdx.x >> 7) * 8, r);
}
template <typename TIn>
__device__ __forceinline__ void stage_load_mc_full(const TIn* __restrict__ src, long ld, int c0, int k0, TIn (&r)[8]) {
  stage_load8(src + (long)(k0 + (threadIdx.x >> 4)) * ld + c0 + (threadIdx.x & 15) * 8, r);
}

// XCD-aware workgroup id: blocks b and b+8 share an XCD (observed round-robin
// dispatch; speed only).  Give every XCD a contiguous run of tiles so that
// neighbouring tiles (same A rows) hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int pid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (pid >> 3);
}

// One 128 x 128 output tile: K loop of the workgroup (staging through LDS, MFMA), result left in the accumulators.
// C/D map of v_mfma_f64_16x16x4_f64 inside the wave's 64 x 64 block: acc[i][j][r] is the element
//   row = wr * 64 + i * 16 + (lane >> 4) + 4 * r,   col = wc * 64 + j * 16 + (lane & 15)      (wr = wave >> 1, wc = wave & 1).
template <typename TA, typename TB, bool A_KC, bool B_KC, bool VEC>
__device__ __forceinline__ void gemm_tile_mainloop(const GemmArgs& g, const TA* __restrict__ A, const TB* __restrict__ B,
                                                   int m0, int n0, int k_lo, int k_hi, double* smem,
                                                   v4f64 (&acc)[4][4]) {
  double* As = smem;                                 // [2][BK][LD]
  double* Bs = smem + 2 * GEMM_BK * GEMM_LD;         // [2][BK][LD]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int kq = lane >> 4, li = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  TA ra[8];
  TB rb[8];
  const int nkt = (k_hi - k_lo + GEMM_BK - 1) / GEMM_BK;

  const bool full_mn = VEC && m0 + GEMM_BM <= g.M && n0 + GEMM_BN <= g.N;  // (uniform over the workgroup)
  auto load_tile = [&](int kt) {
    const int k0 = k_lo + kt * GEMM_BK;
    if (full_mn && k0 + GEMM_BK <= k_hi) {
      if (A_KC) stage_load_kc_full<TA>(A, g.lda, m0, k0, ra); else stage_load_mc_full<TA>(A, g.lda, m0, k0, ra);
      if (B_KC) stage_load_kc_full<TB>(B, g.ldb, n0, k0, rb); else stage_load_mc_full<TB>(B, g.ldb, n0, k0, rb);
      return;
    }
    if (A_KC) stage_load_kc<TA, VEC>(A, g.lda, m0, g.M, k0, k_hi, ra);
    else      stage_load_mc<TA, VEC>(A, g.lda, m0, g.M, k0, k_hi, ra);
    if (B_KC) stage_load_kc<TB, VEC>(B, g.ldb, n0, g.N, k0, k_hi, rb);
    else      stage_load_mc<TB, VEC>(B, g.ldb, n0, g.N, k0, k_hi, rb);
  };
  auto store_tile = [&](int buf) {
    double* a = As + buf * GEMM_BK * GEMM_LD;
    double* b = Bs + buf * GEMM_BK * GEMM_LD;
    if (A_KC) stage_store_kc(a, ra); else stage_store_mc(a, ra);
    if (B_KC) stage_store_kc(b, rb); else stage_store_mc(b, rb);
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    if (more) load_tile(kt + 1);
    const double* a = As + cur * GEMM_BK * GEMM_LD + wr * 64 + li;
    const double* b = Bs + cur * GEMM_BK * GEMM_LD + wc * 64 + li;
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; ++kk) {
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = a[(kk * 4 + kq) * GEMM_LD + i * 16];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = b[(kk * 4 + kq) * GEMM_LD + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
}

// Epilogue concept:  void operator()(int batch, int row, int col, double v) const
template <typename TA, typename TB, bool A_KC, bool B_KC, bool VEC, typename Epi>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f64_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int z = blockIdx.z;
  const int zb = g.bsplit ? z / g.bsplit : z;  // batch entry
  if (g.rep && g.rep[zb] != zb) return;  // a duplicate of entry rep[zb]: its consumer reads that one
  int tm, tn;
  if (g.sym) {
    // C = A A^T with a symmetric epilogue: the grid holds the tiles on or above the diagonal only, row-major over
    // the triangle (consecutive workgroups of an XCD share the A row-panel); row tm starts at
    // e0(tm) = tm * tiles - tm (tm - 1) / 2.  Equal tile counts per XCD (a full grid with early exits gives the
    // XCD that owns the first tile-rows twice the average work).
    const int t = g.tiles_n, nwg = t * (t + 1) / 2;
    const int e = xcd_remap(blockIdx.x, nwg);
    int r = (int)((2.0 * t + 1.0 - sqrt((2.0 * t + 1.0) * (2.0 * t + 1.0) - 8.0 * e)) * 0.5);
    r = r < 0 ? 0 : (r > t - 1 ? t - 1 : r);
    while (r > 0 && r * t - r * (r - 1) / 2 > e) --r;
    while (r + 1 < t && (r + 1) * t - (r + 1) * r / 2 <= e) ++r;
    tm = r;
    tn = r + (e - (r * t - r * (r - 1) / 2));
  } else {
    const int nwg = g.tiles_m * g.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    // grouped tile order: consecutive workgroup ids (one XCD, dispatched together) cover 8 tile-rows x 8
    // tile-columns, so that the ~64 tiles in flight on an XCD re-use 8 A row-panels and 8 B column-panels
    // out of its L2 instead of streaming 64 different B panels from the Infinity Cache
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * g.tiles_n;
    const int group_id = wg / per_group;
    const int first_m = group_id * GROUP_M;
    const int group_size = min(g.tiles_m - first_m, GROUP_M);
    tm = first_m + (wg % group_size);
    tn = (wg % per_group) / group_size;
  }
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
  const bool mirror = g.sym && tm != tn;

  int k_lo = 0, k_hi = g.K;
  const TA* A = reinterpret_cast<const TA*>(g.A);
  const TB* B = reinterpret_cast<const TB*>(g.B);
  if (g.splitk) {
    k_lo = z * g.kchunk;
    k_hi = min(g.K, k_lo + g.kchunk);
  } else {
    A += (long)zb * g.strideA;
    B += (long)zb * g.strideB;
    if (g.bsplit) {
      k_lo = (z - zb * g.bsplit) * g.kchunk;
      k_hi = min(g.K, k_lo + g.kchunk);
    }
  }

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int kq = lane >> 4, li = lane & 15;

  v4f64 acc[4][4];
  gemm_tile_mainloop<TA, TB, A_KC, B_KC, VEC>(g, A, B, m0, n0, k_lo, k_hi, smem, acc);

  // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 64 + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + i * 16 + kq + 4 * r;
        if (row < g.M && col < g.N) epi(z, row, col, acc[i][j][r]);
      }
    }
  }
  if (mirror) {
    // the same 64 x 64 block of the wave, written to (col, row): transposed through a wave-private LDS patch
    // (32 columns at a time, pitch 65) so that every store instruction writes 512 contiguous bytes of one row
    __syncthreads();  // all waves are done with the operand buffers
    double* patch = smem + wave * (32 * 65);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = half * 2 + jj;
          const int col = n0 + wc * 64 + j * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = i * 16 + kq + 4 * r;
            // entry (col, row) of the result: evaluated in ITS orientation (the epilogue need not be commutative)
            const int rg = m0 + wr * 64 + rl;
            patch[(jj * 16 + li) * 65 + rl] = (rg < g.M && col < g.N) ? epi.value(col, rg, acc[i][j][r]) : 0.0;
          }
        }
      // wave-private patch: LDS operations of one wave complete in order, no barrier
      const int rowm = m0 + wr * 64 + lane;
#pragma unroll 8
      for (int c = 0; c < 32; ++c) {
        const int colm = n0 + wc * 64 + half * 32 + c;
        if (colm < g.N && rowm < g.M) epi.put(z, colm, rowm, patch[c * 65 + lane]);
      }
    }
  }
}

// ---- plain epilogue --------------------------------------------------------
struct EpiStore {
  double* C;
  long ldc;
  long strideC;
  double alpha;
  __device__ __forceinline__ void operator()(int z, int row, int col, double v) const {
    C[(long)z * strideC + (long)row * ldc + col] = alpha * v;
  }
  // symmetric launches: value(row, col, v) is what operator() would store, put() stores it somewhere else
  __device__ __forceinline__ double value(int, int, double v) const { return alpha * v; }
  __device__ __forceinline__ void put(int z, int row, int col, double val) const {
    C[(long)z * strideC + (long)row * ldc + col] = val;
  }
};

template <typename TA, typename TB, bool A_KC, bool B_KC, bool VEC, typename Epi>
int gemm_f64_prepare_t() {
  // > 64 KiB of dynamic LDS needs the attribute once per kernel; done outside stream capture.  Several host
  // threads drive the library (sketch groups + the main path): one-time, thread-safe.
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] {
    auto k = gemm_f64_kernel<TA, TB, A_KC, B_KC, VEC, Epi>;
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              GEMM_LDS_BYTES);
  });
  MUSED_CHECK_HIP(err);
  return MUSED_OK;
}

template <typename TA, typename TB, bool A_KC, bool B_KC, typename Epi>
int gemm_f64_launch_t(GemmArgs g, int batch, Epi epi, bool vec, hipStream_t stream) {
  g.tiles_m = cdiv(g.M, GEMM_BM);
  g.tiles_n = cdiv(g.N, GEMM_BN);
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return MUSED_OK;
  dim3 grid(g.sym ? g.tiles_n * (g.tiles_n + 1) / 2 : g.tiles_m * g.tiles_n, 1, batch);
  int rc;
  if (vec) {
    if ((rc = gemm_f64_prepare_t<TA, TB, A_KC, B_KC, true, Epi>())) return rc;
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, A_KC, B_KC, true, Epi>), grid, dim3(GEMM_THREADS), GEMM_LDS_BYTES,
                       stream, g, epi);
  } else {
    if ((rc = gemm_f64_prepare_t<TA, TB, A_KC, B_KC, false, Epi>())) return rc;
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, A_KC, B_KC, false, Epi>), grid, dim3(GEMM_THREADS), GEMM_LDS_BYTES,
                       stream, g, epi);
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// 16-byte vector loads are legal when base pointers, leading dimensions and
// batch strides keep every 8-element group 16-B aligned.
template <typename T>
static inline bool vec_ok(const void* p, long ld, long stride) {
  const long e = 16 / (long)sizeof(T);
  return ((uintptr_t)p % 16 == 0) && (ld % e == 0) && (stride % e == 0);
}

}  // namespace mused
