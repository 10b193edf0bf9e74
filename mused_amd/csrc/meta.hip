// Metadata modalities of create_adjacency_matrix (/root/reference/matrix_operations.py:22-89): score matrices for the
// two-column record types ("location": haversine km, :22-31 + :250-263; "time": |d datetaken| + |d dateupload|,
// :33-54), the Jaccard similarity of tag sets ("tags", :73-89 + :245-248) and the same-user relation ("username",
// :56-71).  The k smallest scores per row are then taken by mused_select_k_smallest (ties to the smaller column; the
// reference's own tie order is that of an unstable sort / a tree traversal and is not defined).  All of this is small
// HBM-bound work next to the dense modalities: one pass that writes n x n scores (or n x n / 64 mask words).
#include <mutex>

#include "common.h"
#include "internal.h"
#include "meta_scores.h"

// no fused multiply-add anywhere in this file: the scores follow the host expressions operation by operation
#pragma clang fp contract(off)

namespace mused {

// ---- two-column records: haversine_km / time_l1 live in meta_scores.h (shared with the selection kernel of knn.hip) ---

template <int KIND>
__global__ __launch_bounds__(256) void record_scores_kernel(const double* __restrict__ rec, int n, double* __restrict__ S) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 8;
  if (j >= n) return;
  const double bj0 = rec[2 * (long)j], bj1 = rec[2 * (long)j + 1];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = i0 + r;
    if (i >= n) break;
    const double a0 = rec[2 * (long)i], a1 = rec[2 * (long)i + 1];
    double s;
    if (KIND == 0) {
      // the query row is location1, the candidate location2 (the formula is symmetric in them)
      s = haversine_km(a0, a1, bj0, bj1);
    } else {
      s = time_l1(a0, a1, bj0, bj1);
    }
    S[(long)i * n + j] = s;
  }
}

// ---- username: A[i][j] = 1 iff same (non-negative) id and i != j ----------------------------------------------------
__global__ __launch_bounds__(256) void group_mask_kernel(const int* __restrict__ ids, int n, int words,
                                                         unsigned long long* __restrict__ mask) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (w >= words) return;
  const int me = ids[i];
  unsigned long long bits = 0;
  if (me >= 0) {
    const int j0 = w * 64;
    for (int b = 0; b < 64 && j0 + b < n; ++b)
      if (ids[j0 + b] == me && j0 + b != i) bits |= 1ull << b;
  }
  mask[(long)i * words + w] = bits;
}

// ---- tags: S[i][j] = -|T_i & T_j| / |T_i | T_j| (0 when either set is empty), S[i][i] = +1 ---------------------------
// One workgroup per row i.  The intersection sizes with every other row are counted in LDS by walking the posting
// lists of row i's tags: rows inside one posting list are distinct, so a list is added without atomics and the lists
// are separated by barriers.
__global__ __launch_bounds__(256) void jaccard_scores_kernel(const int* __restrict__ rowptr, const int* __restrict__ tags,
                                                             const int* __restrict__ postptr,
                                                             const int* __restrict__ postrow, int n,
                                                             double* __restrict__ S) {
  extern __shared__ unsigned short inter[];
  const int i = blockIdx.x;
  for (int j = threadIdx.x; j < n; j += 256) inter[j] = 0;
  __syncthreads();
  const int t0 = rowptr[i], t1 = rowptr[i + 1];
  for (int t = t0; t < t1; ++t) {
    const int tag = tags[t];
    const int p0 = postptr[tag], p1 = postptr[tag + 1];
    for (int p = p0 + threadIdx.x; p < p1; p += 256) inter[postrow[p]] += 1;
    __syncthreads();
  }
  const int li = t1 - t0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const int lj = rowptr[j + 1] - rowptr[j];
    double s = 0.0;
    if (li > 0 && lj > 0) {
      const int in = inter[j];
      s = 0.0 - (double)in / (double)(li + lj - in);  // true division of two ints, then negated: -0.0 never stored
    }
    S[(long)i * n + j] = j == i ? 1.0 : s;
  }
}

}  // namespace mused

using namespace mused;

extern "C" {

int mused_record_scores(const double* rec, int n, int kind, double* S, void* stream) {
  MUSED_REQUIRE(rec && S && n > 0 && (kind == 0 || kind == 1), "mused_record_scores: bad arguments (n=%d kind=%d)", n, kind);
  dim3 grid(cdiv(n, 256), cdiv(n, 8));
  if (kind == 0)
    record_scores_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(rec, n, S);
  else
    record_scores_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(rec, n, S);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int mused_group_mask(const int* ids, int n, unsigned long long* out_mask, int mask_words, void* stream) {
  MUSED_REQUIRE(ids && out_mask && n > 0 && mask_words >= cdiv(n, 64) && n <= 65535,
                "mused_group_mask: bad arguments (n=%d words=%d)", n, mask_words);
  dim3 grid(cdiv(mask_words, 256), n);
  group_mask_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(ids, n, mask_words, out_mask);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int mused_jaccard_scores(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int n_tags,
                         double* S, void* stream) {
  MUSED_REQUIRE(rowptr && postptr && S && n > 0 && n_tags >= 0, "mused_jaccard_scores: bad arguments (n=%d)", n);
  // a set has < 65536 tags (16-bit intersection counters) and the counters of one row fit the 160 KB of LDS
  MUSED_REQUIRE(n <= 65536, "mused_jaccard_scores: at most 65536 rows per window (n=%d)", n);
  const size_t lds = (size_t)n * sizeof(unsigned short);
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)jaccard_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  128 * 1024);
  });
  MUSED_CHECK_HIP(attr_rc);
  jaccard_scores_kernel<<<n, 256, lds, (hipStream_t)stream>>>(rowptr, tags, postptr, postrow, n, S);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// The same selections WITHOUT the n x n score matrix (n <= 16384: the scores of a row live in LDS only): the k closest
// rows per row by haversine / time distance, resp. the k rows of largest Jaccard similarity; ties to the smaller row;
// outputs as mused_select_k_smallest.
int mused_record_knn(const double* rec, int n, int kind, int k, int* out_idx, unsigned long long* out_mask, int mask_words,
                     void* stream) {
  MUSED_REQUIRE(rec && n > 0 && (kind == 0 || kind == 1) && k >= 1 && k <= n && n <= select_max_fused_rows(false),
                "mused_record_knn: bad arguments (n=%d kind=%d k=%d)", n, kind, k);
  return select_from_records(rec, n, kind, k, out_idx, out_mask, mask_words, (hipStream_t)stream);
}

int mused_jaccard_knn(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int n_tags, int k,
                      int* out_idx, unsigned long long* out_mask, int mask_words, void* stream) {
  MUSED_REQUIRE(rowptr && postptr && n > 0 && n_tags >= 0 && k >= 1 && k <= n && n <= select_max_fused_rows(true),
                "mused_jaccard_knn: bad arguments (n=%d k=%d)", n, k);
  return select_from_tag_sets(rowptr, tags, postptr, postrow, n, k, out_idx, out_mask, mask_words, (hipStream_t)stream);
}

}  // extern "C"
