// a1/a2 of SURVEY section 8: pairwise scores (squared-L2 or negated cosine) by fp64
// MFMA GEMM with a fused epilogue, then per-row selection of the k smallest scores.
//
// Reference behaviour reproduced (matrix_operations.py:112-119 via sklearn's
// EuclideanArgKmin; :101-108 via cosine_similarity + argsort):
//   l2     : score(i,j) = max(0, |x_i|^2 - 2 x_i.x_j + |x_j|^2), fp64
//   cosine : score(i,j) = -(x_i.x_j) / (max(|x_i|,!0) * max(|x_j|,!0)), fp64
//   neighbours of i = the k rows of smallest score, ties at the k-th value towards
//   the smaller index (sklearn's heap keeps the earlier candidate on ties).
#include "gemm_f64.h"
#include "internal.h"
#include "meta_scores.h"

namespace mused {

// ---- row squared norms (HBM-bound: one wave per row, 16-B loads) ------------
template <typename T>
__global__ void row_sqnorm_kernel(const T* __restrict__ X, long n, int d, long ld, double* __restrict__ out) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const T* p = X + row * ld;
  double s = 0.0;
  constexpr int V = 16 / sizeof(T);
  const bool vec = ((uintptr_t)p % 16 == 0);
  int c = 0;
  if (vec) {
    for (c = lane * V; c + V <= d; c += 64 * V) {
      if (sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      } else {
        const double2 v = *reinterpret_cast<const double2*>(p + c);
        s += v.x * v.x + v.y * v.y;
      }
    }
    // tail columns (d not a multiple of V): handled below from the first uncovered column
    c = (d / V) * V;
    for (int t = c + lane; t < d; t += 64) s += (double)p[t] * (double)p[t];
  } else {
    for (int t = lane; t < d; t += 64) s += (double)p[t] * (double)p[t];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ void inv_norm_kernel(const double* __restrict__ sq, long n, double* __restrict__ inv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double nr = sqrt(sq[i]);
  inv[i] = (nr == 0.0) ? 1.0 : 1.0 / nr;  // sklearn normalize(): zero norms replaced by 1
}

int inv_norms_launch(double* norms, long n, hipStream_t st) {
  hipLaunchKernelGGL(inv_norm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, norms, n, norms);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// ---- epilogues ---------------------------------------------------------------
struct EpiSqL2 {
  double* S;
  long ld;
  const double* nrm;  // squared norms
  __device__ __forceinline__ void operator()(int, int row, int col, double v) const {
    S[(long)row * ld + col] = value(row, col, v);
  }
  __device__ __forceinline__ double value(int row, int col, double v) const {
    const double dd = nrm[row] - 2.0 * v + nrm[col];
    return dd > 0.0 ? dd : 0.0;
  }
  __device__ __forceinline__ void put(int, int row, int col, double val) const { S[(long)row * ld + col] = val; }
};

struct EpiNegCos {
  double* S;
  long ld;
  const double* inv;  // 1 / norm
  __device__ __forceinline__ void operator()(int, int row, int col, double v) const {
    S[(long)row * ld + col] = value(row, col, v);
  }
  // 0.0 - x, not -x: a zero dot product gives +0.0 whatever its sign, so that equal similarities are equal KEYS for the
  // selection too (ties then go to the smaller column, as everywhere else)
  __device__ __forceinline__ double value(int row, int col, double v) const { return 0.0 - (v * inv[row] * inv[col]); }
  __device__ __forceinline__ void put(int, int row, int col, double val) const { S[(long)row * ld + col] = val; }
};

template <typename T>
static int scores_launch(const T* X, long n, int d, long ld, int metric, double* norms, double* S,
                         hipStream_t stream) {
  hipLaunchKernelGGL(row_sqnorm_kernel<T>, dim3(cdiv(n, 4)), dim3(256), 0, stream, X, n, d, ld, norms);
  MUSED_LAUNCH_CHECK();
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = X; g.B = X; g.lda = ld; g.ldb = ld; g.M = (int)n; g.N = (int)n; g.K = d;
  // X X^T: the squared-distance / cosine epilogues are symmetric in (row, col) -> upper tiles only, mirrored stores
  static const int sym = [] {
    const char* sy = getenv("MUSED_SCORES_SYM");
    return (sy && sy[0] == '0') ? 0 : 1;
  }();
  g.sym = sym;
  const bool vec = vec_ok<T>(X, ld, 0);
  if (metric == 0) {
    EpiSqL2 epi{S, n, norms};
    return gemm_f64_launch_t<T, T, true, true>(g, 1, epi, vec, stream);
  }
  hipLaunchKernelGGL(inv_norm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, norms, n, norms);
  MUSED_LAUNCH_CHECK();
  EpiNegCos epi{S, n, norms};
  return gemm_f64_launch_t<T, T, true, true>(g, 1, epi, vec, stream);
}

// ---- k-smallest selection -------------------------------------------------------
// One 1024-thread workgroup per row.  Exact radix select on order-preserving 64-bit
// keys: bits above the highest bit in which the row's min and max keys differ are
// skipped; each pass histograms one digit (<= 8 bits) of the still-undecided keys
// in LDS; when the bin holding the k-th key has shrunk to <= 256 keys the rest is
// finished by ranking those candidates against each other.  Selected columns are
// emitted in ascending column order (block prefix sums) -> deterministic output.
constexpr int SEL_THREADS = 1024;
constexpr int SEL_WAVES = SEL_THREADS / 64;
constexpr int SEL_MAX_LDS_KEYS = 16384;  // 128 KiB of keys cached in LDS; longer rows re-read global
constexpr int SEL_MAX_JACCARD_ROWS = 15000;  // keys + 16-bit intersection counters of one row: 150 KB of the 160 KB

__device__ __forceinline__ int block_excl_scan(int v, int* ws /*[SEL_WAVES]*/, int& total) {
  // exclusive prefix sum over the workgroup; ws in LDS
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) ws[w] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SEL_WAVES; ++i) {
    const int c = ws[i];
    base += (i < w) ? c : 0;
    tot += c;
  }
  total = tot;
  __syncthreads();
  return base + x - v;
}

// Where a row of scores comes from.  SRC_MATRIX: row `row` of S (the classic path).  The others compute the row on the
// fly (rows of at most SEL_MAX_LDS_KEYS scores, kept as keys in LDS) -- the metadata modality types of meta.hip without
// an n x n score matrix: SRC_HAVERSINE / SRC_TIME from n x 2 records, SRC_JACCARD from tag sets (CSR + posting lists).
enum { SRC_MATRIX = 0, SRC_HAVERSINE = 1, SRC_TIME = 2, SRC_JACCARD = 3 };
struct SelSource {
  const double* rec;     // SRC_HAVERSINE / SRC_TIME: n x 2 records
  const int* rowptr;     // SRC_JACCARD: tag sets as CSR ...
  const int* tags;
  const int* postptr;    // ... and the posting lists of the tags
  const int* postrow;
};

template <int SRC>
__global__ __launch_bounds__(SEL_THREADS) void select_k_kernel(const double* __restrict__ S, long ld, int n, int k,
                                                             int* __restrict__ out_idx /* n x k or null */,
                                                             unsigned long long* __restrict__ out_mask, int mask_words,
                                                             SelSource src) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sel_smem[];
  __shared__ int hist[256];
  __shared__ int ws[SEL_WAVES];
  __shared__ unsigned long long s_red[2 * SEL_WAVES];
  __shared__ unsigned long long s_prefix, s_maskbits;
  __shared__ int s_remaining, s_shift, s_done, s_ncand, s_eqcount;
  __shared__ unsigned long long cand_key[256];
  __shared__ int cand_idx[256];

  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const double* srow = SRC == SRC_MATRIX ? S + (long)row * ld : nullptr;
  const bool cached = (SRC != SRC_MATRIX) || (n <= SEL_MAX_LDS_KEYS);
  unsigned long long* keys = sel_smem;  // [n] when cached

  [[maybe_unused]] unsigned short* inter = reinterpret_cast<unsigned short*>(sel_smem + n);  // SRC_JACCARD: behind the keys
  [[maybe_unused]] double r0 = 0.0, r1 = 0.0;
  [[maybe_unused]] int li = 0;
  if constexpr (SRC == SRC_HAVERSINE || SRC == SRC_TIME) {
    r0 = src.rec[2 * (long)row];
    r1 = src.rec[2 * (long)row + 1];
  }
  if constexpr (SRC == SRC_JACCARD) {
    // intersection sizes of this row's tag set with every row's: walk the posting lists of its tags (rows inside a
    // list are distinct: no atomics, a barrier between lists) -- jaccard_scores_kernel of meta.hip
    for (int j = tid; j < n; j += SEL_THREADS) inter[j] = 0;
    __syncthreads();
    const int t0 = src.rowptr[row], t1 = src.rowptr[row + 1];
    li = t1 - t0;
    for (int t = t0; t < t1; ++t) {
      const int tag = src.tags[t];
      const int p1 = src.postptr[tag + 1];
      for (int q = src.postptr[tag] + tid; q < p1; q += SEL_THREADS) inter[src.postrow[q]] += 1;
      __syncthreads();
    }
  }
  auto score_at = [&](int i) -> double {
    if constexpr (SRC == SRC_MATRIX) return srow[i];
    if constexpr (SRC == SRC_HAVERSINE) return haversine_km(r0, r1, src.rec[2 * (long)i], src.rec[2 * (long)i + 1]);
    if constexpr (SRC == SRC_TIME) return time_l1(r0, r1, src.rec[2 * (long)i], src.rec[2 * (long)i + 1]);
    if constexpr (SRC == SRC_JACCARD) {
      if (i == row) return 1.0;
      const int lj = src.rowptr[i + 1] - src.rowptr[i];
      if (li == 0 || lj == 0) return 0.0;
      const int in = inter[i];
      return 0.0 - (double)in / (double)(li + lj - in);
    }
    return 0.0;
  };

  // pass 0: keys, min, max
  unsigned long long kmin = ~0ull, kmax = 0ull;
  for (int i = tid; i < n; i += SEL_THREADS) {
    const unsigned long long key = f64_key(score_at(i));
    if (cached) keys[i] = key;
    kmin = key < kmin ? key : kmin;
    kmax = key > kmax ? key : kmax;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  if ((tid & 63) == 0) {
    s_red[tid >> 6] = kmin;
    s_red[SEL_WAVES + (tid >> 6)] = kmax;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long mn = s_red[0], mx = s_red[SEL_WAVES];
    for (int i = 1; i < SEL_WAVES; ++i) {
      mn = s_red[i] < mn ? s_red[i] : mn;
      mx = s_red[SEL_WAVES + i] > mx ? s_red[SEL_WAVES + i] : mx;
    }
    const unsigned long long diff = mn ^ mx;
    s_remaining = k;
    if (diff == 0) {  // all keys equal: threshold = that key, every needed element is "equal"
      s_prefix = mn;
      s_maskbits = ~0ull;
      s_shift = -1;
    } else {
      const int hb = 63 - __clzll(diff);       // highest differing bit
      const int top = hb + 1;                  // bits [top, 64) are common
      s_maskbits = (top >= 64) ? 0ull : (~0ull << top);
      s_prefix = mn & s_maskbits;
      s_shift = top;                           // next digit covers [max(0, top-8), top)
    }
    s_done = 0;
    s_ncand = 0;
  }
  __syncthreads();

  auto key_at = [&](int i) -> unsigned long long { return cached ? keys[i] : f64_key(srow[i]); };

  // radix passes
  while (true) {
    const int top = s_shift;
    if (top <= 0 || s_done) break;
    const int lo = top >= 8 ? top - 8 : 0;
    const int width = top - lo;
    const unsigned long long pmask = s_maskbits, prefix = s_prefix;
    for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SEL_THREADS) {
      const unsigned long long key = key_at(i);
      if ((key & pmask) == prefix) atomicAdd(&hist[(int)((key >> lo) & ((1u << width) - 1))], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining, cum = 0, dsel = 0;
      const int nb = 1 << width;
      for (int b = 0; b < nb; ++b) {
        if (cum + hist[b] >= rem) { dsel = b; break; }
        cum += hist[b];
      }
      s_remaining = rem - cum;
      s_prefix = prefix | ((unsigned long long)dsel << lo);
      s_maskbits = pmask | ((((unsigned long long)1 << width) - 1) << lo);
      s_shift = lo;
      if (hist[dsel] <= 256 && lo > 0) s_done = 1;  // finish by ranking the bin's members
    }
    __syncthreads();
  }

  unsigned long long thr;  // k-th smallest key
  int need_eq;             // how many keys == thr are selected (smallest indices first)
  if (s_done) {
    // gather the undecided bin (<= 256 keys) with an LDS atomic append; the order does not matter,
    // the ranking below is by (key, column)
    const unsigned long long pmask = s_maskbits, prefix = s_prefix;
    for (int i = tid; i < n; i += SEL_THREADS) {
      const unsigned long long key = key_at(i);
      if ((key & pmask) == prefix) {
        const int pos = atomicAdd(&s_ncand, 1);
        cand_key[pos] = key;
        cand_idx[pos] = i;
      }
    }
    __syncthreads();
    const int nc = s_ncand, rem = s_remaining;  // the rem-th smallest (1-based) of the candidates
    if (tid < nc) {
      const unsigned long long mk = cand_key[tid];
      const int mi = cand_idx[tid];
      int rank = 0, eq_before = 0, eq_all = 0;
      for (int j = 0; j < nc; ++j) {
        const unsigned long long kj = cand_key[j];
        const bool same = (kj == mk);
        const bool before = same && cand_idx[j] < mi;
        rank += (kj < mk) || before;
        eq_before += before;
        eq_all += same;
      }
      if (rank == rem - 1) {
        s_prefix = mk;
        s_remaining = eq_before + 1;
        s_eqcount = eq_all;
      }
    }
    __syncthreads();
  } else if (tid == 0) {
    s_eqcount = -1;  // unknown: the general emit path counts
  }
  __syncthreads();
  thr = s_prefix;
  need_eq = s_remaining;

  // emit: keys < thr, plus the first need_eq keys == thr in index order
  unsigned long long* mrow = out_mask ? out_mask + (long)row * mask_words : nullptr;
  // LDS bitmask staging reuses hist/cand arrays is awkward for large n; write mask words with
  // wave ballots instead: each group of 64 consecutive columns is one word owned by one wave.
  if (!out_idx && s_eqcount == need_eq) {
    // common case: every key equal to the threshold is selected (no tie is cut) and only the bitmask
    // is wanted -> one ballot per 64 columns, no prefix sums, no barriers
    for (int base = 0; base < n; base += SEL_THREADS) {
      const int i = base + tid;
      const bool sel = (i < n) && (key_at(i) <= thr) && (i != row);
      const unsigned long long bal = __ballot(sel);
      const int word = i >> 6;
      if ((tid & 63) == 0 && word < mask_words) mrow[word] = bal;
    }
    const int first_free = ((n + SEL_THREADS - 1) / SEL_THREADS) * (SEL_THREADS / 64);
    for (int w = first_free + tid; w < mask_words; w += SEL_THREADS) mrow[w] = 0ull;
    return;
  }
  int eq_seen = 0, emitted = 0;
  for (int base = 0; base < n; base += SEL_THREADS) {
    const int i = base + tid;
    unsigned long long key = ~0ull;
    int is_lt = 0, is_eq = 0;
    if (i < n) {
      key = key_at(i);
      is_lt = key < thr;
      is_eq = key == thr;
    }
    int tot_eq;
    const int eq_pos = block_excl_scan(is_eq, ws, tot_eq);
    const int sel = is_lt || (is_eq && (eq_seen + eq_pos) < need_eq);
    int tot_sel;
    const int pos = block_excl_scan(sel, ws, tot_sel);
    if (sel && out_idx) out_idx[(long)row * k + emitted + pos] = i;
    if (mrow) {
      const unsigned long long bal = __ballot(sel && i != row);  // self edge dropped (matrix_operations.py:128)
      const int word = i >> 6;  // base is a multiple of 64 -> each wave covers exactly one word
      if ((tid & 63) == 0 && word < mask_words) mrow[word] = bal;
    }
    eq_seen += tot_eq;
    emitted += tot_sel;
  }
  // zero any mask words past the last processed column group
  if (mrow) {
    const int first_free = ((n + SEL_THREADS - 1) / SEL_THREADS) * (SEL_THREADS / 64);
    for (int w = first_free + tid; w < mask_words; w += SEL_THREADS) mrow[w] = 0ull;
  }
}

template <int SRC>
static int select_launch_src(const double* S, long ld, int n, int k, int* out_idx, unsigned long long* out_mask,
                             int mask_words, const SelSource& src, hipStream_t stream) {
  // keys of the row (cached case) + the 16-bit intersection counters of SRC_JACCARD behind them
  const size_t lds = ((SRC != SRC_MATRIX || n <= SEL_MAX_LDS_KEYS) ? (size_t)n * 8 : 0) + (SRC == SRC_JACCARD ? (size_t)n * 2 + 8 : 0);
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(select_k_kernel<SRC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SRC == SRC_JACCARD ? SEL_MAX_JACCARD_ROWS * 10 + 64 : SEL_MAX_LDS_KEYS * 8);
  });
  MUSED_CHECK_HIP(attr_rc);
  hipLaunchKernelGGL(select_k_kernel<SRC>, dim3(n), dim3(SEL_THREADS), lds, stream, S, ld, n, k, out_idx, out_mask,
                     mask_words, src);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

static int select_launch(const double* S, long ld, int n, int k, int* out_idx, unsigned long long* out_mask,
                         int mask_words, hipStream_t stream) {
  return select_launch_src<SRC_MATRIX>(S, ld, n, k, out_idx, out_mask, mask_words, SelSource{}, stream);
}

// selection straight from the records / tag sets (rows of at most SEL_MAX_LDS_KEYS scores): meta.hip's C entry points
int select_from_records(const double* rec, int n, int kind, int k, int* out_idx, unsigned long long* out_mask,
                        int mask_words, hipStream_t stream) {
  SelSource src{};
  src.rec = rec;
  if (kind == 0) return select_launch_src<SRC_HAVERSINE>(nullptr, 0, n, k, out_idx, out_mask, mask_words, src, stream);
  return select_launch_src<SRC_TIME>(nullptr, 0, n, k, out_idx, out_mask, mask_words, src, stream);
}
int select_from_tag_sets(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int k,
                         int* out_idx, unsigned long long* out_mask, int mask_words, hipStream_t stream) {
  SelSource src{};
  src.rowptr = rowptr; src.tags = tags; src.postptr = postptr; src.postrow = postrow;
  return select_launch_src<SRC_JACCARD>(nullptr, 0, n, k, out_idx, out_mask, mask_words, src, stream);
}
int select_max_fused_rows(bool tag_sets) { return tag_sets ? SEL_MAX_JACCARD_ROWS : SEL_MAX_LDS_KEYS; }

}  // namespace mused

using namespace mused;

extern "C" {

// Replaces: `np.linalg.norm(...)**2` style row norms (main.py:61) and the norm stage of
// sklearn's Euclidean/cosine kernels.  out[i] = sum_j X[i,j]^2 in fp64.
int mused_row_sq_norms(const void* X, int dtype, long n, int d, long ld, double* out, void* stream) {
  MUSED_REQUIRE(X && out && n >= 0 && d > 0 && ld >= d, "mused_row_sq_norms: bad arguments");
  if (n == 0) return MUSED_OK;
  if (dtype == MUSED_F32)
    hipLaunchKernelGGL(row_sqnorm_kernel<float>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)X, n, d, ld, out);
  else if (dtype == MUSED_F64)
    hipLaunchKernelGGL(row_sqnorm_kernel<double>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const double*)X, n, d, ld, out);
  else {
    set_error("mused_row_sq_norms: unsupported dtype %d", dtype);
    return MUSED_ERR_UNSUPPORTED;
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// Pairwise score matrix S (n x n fp64, ld = n).  metric 0 = squared L2, 1 = negated cosine.
// `norms` is an n-double workspace (squared norms for l2, 1/norm for cosine on return).
int mused_pairwise_scores(const void* X, int dtype, long n, int d, long ld, int metric, double* norms, double* S,
                          void* stream) {
  MUSED_REQUIRE(X && norms && S && n > 0 && d > 0 && ld >= d, "mused_pairwise_scores: bad arguments");
  MUSED_REQUIRE(metric == 0 || metric == 1, "mused_pairwise_scores: metric must be 0 (l2) or 1 (cosine)");
  MUSED_REQUIRE(n < (1l << 31), "mused_pairwise_scores: n too large");
  if (dtype == MUSED_F32) return scores_launch<float>((const float*)X, n, d, ld, metric, norms, S, (hipStream_t)stream);
  if (dtype == MUSED_F64) return scores_launch<double>((const double*)X, n, d, ld, metric, norms, S, (hipStream_t)stream);
  set_error("mused_pairwise_scores: unsupported dtype %d", dtype);
  return MUSED_ERR_UNSUPPORTED;
}

// Per row, the k smallest entries of S (ties -> smaller column).  out_idx (n x k int32,
// ascending column order) and/or out_mask (n x mask_words uint64 bitmask with the row's
// own column cleared) may be null.
int mused_select_k_smallest(const double* S, long ld, int n, int k, int* out_idx, unsigned long long* out_mask,
                            int mask_words, void* stream) {
  MUSED_REQUIRE(S && n > 0 && k >= 1 && k <= n && ld >= n, "mused_select_k_smallest: need 1 <= k <= n (k=%d n=%d)", k, n);
  MUSED_REQUIRE(!out_mask || mask_words >= (n + 63) / 64, "mused_select_k_smallest: mask_words too small");
  return select_launch(S, ld, n, k, out_idx, out_mask, mask_words, (hipStream_t)stream);
}

// Replaces sklearn NearestNeighbors(...).kneighbors (matrix_operations.py:118-119) and the
// cosine + argsort top-k of :106-108 for dense rows: scores + selection in one call.
// ws_scores: n*n doubles, ws_norms: n doubles (caller-allocated workspaces).
int mused_knn_topk(const void* X, int dtype, long n, int d, long ld, int k, int metric, double* ws_scores,
                   double* ws_norms, int* out_idx, unsigned long long* out_mask, int mask_words, void* stream) {
  int rc = mused_pairwise_scores(X, dtype, n, d, ld, metric, ws_norms, ws_scores, stream);
  if (rc) return rc;
  return mused_select_k_smallest(ws_scores, n, (int)n, k, out_idx, out_mask, mask_words, stream);
}

}  // extern "C"
