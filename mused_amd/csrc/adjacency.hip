// a3/a4 of SURVEY section 8 and the device-resident adjacency representation.
//
// The reference materialises every adjacency as a dense n x n float64/int64 matrix
// (matrix_operations.py:17, 134-141; 800 MB at n = 10^4).  On the device an adjacency is
// an n x words uint64 BITMASK (bit j of row i set <=> A[i, j] = 1; words = ceil(n/64),
// row pitch `words`), 12.5 MB at n = 10^4.  Fusion (logical OR) is a word-wise OR, R of
// main.py:61 is the largest row popcount, and the eigenstep consumes CSR neighbour
// lists derived from the bitmask (rows in ascending column order -> reproducible sums).
// Every kernel here is HBM-bound integer/bit work.
#include "internal.h"

namespace mused {

// out = OR_m masks[m]   (matrix_operations.py:134-141)
__global__ void mask_or_kernel(const unsigned long long* const* __restrict__ masks, int M, long total,
                               unsigned long long* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  unsigned long long v = 0;
  for (int m = 0; m < M; ++m) v |= masks[m][i];
  out[i] = v;
}

__global__ void mask_or2_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b,
                                long total, unsigned long long* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = a[i] | b[i];
}

// deg[i] = popcount(row i); one wave per row
__global__ void mask_degree_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                   int* __restrict__ deg) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  int c = 0;
  for (int w = lane; w < words; w += 64) c += __popcll(mask[(long)row * words + w]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) deg[row] = c;
}

// single workgroup: rowptr = exclusive scan(deg), stats[0] = max degree, stats[1] = nnz.
// cap > 0: every rowptr entry is clamped to cap (the capacity of the colidx array the lists go to) and *overflow
// is raised when nnz > cap, so that a consumer walking rowptr[i] .. rowptr[i+1] never leaves colidx -- the lists
// are then truncated (results invalid, flagged), never out of bounds.
__global__ __launch_bounds__(1024) void degree_scan_kernel(const int* __restrict__ deg, int n, int* __restrict__ rowptr,
                                                          int* __restrict__ stats, long cap = 0,
                                                          int* __restrict__ overflow = nullptr) {
  __shared__ int wsum[16];
  __shared__ int carry;
  __shared__ int smax;
  if (threadIdx.x == 0) { carry = 0; smax = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < n) ? deg[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    int m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (lane == 0) atomicMax(&smax, m);
    __syncthreads();
    int b = carry, tot = 0;
    for (int j = 0; j < 16; ++j) {
      const int c = wsum[j];
      b += (j < w) ? c : 0;
      tot += c;
    }
    if (i < n) {
      const int rp = b + x - v;
      rowptr[i] = (cap > 0 && rp > cap) ? (int)cap : rp;
    }
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rowptr[n] = (cap > 0 && carry > cap) ? (int)cap : carry;
    stats[0] = smax;
    stats[1] = carry;
    if (cap > 0 && carry > cap && overflow) atomicOr(overflow, 1);
  }
}

// colidx[rowptr[i] ...] = set bits of row i in ascending order; one wave per row
__global__ void mask_to_csr_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                   const int* __restrict__ rowptr, int* __restrict__ colidx, long cap,
                                   int* __restrict__ overflow) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  int base = rowptr[row];
  for (int w0 = 0; w0 < words; w0 += 64) {
    const int w = w0 + lane;
    unsigned long long bits = (w < words) ? mask[(long)row * words + w] : 0ull;
    const int c = __popcll(bits);
    int x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    int pos = base + x - c;
    while (bits) {
      const int b = __ffsll((long long)bits) - 1;
      if (cap > 0 && pos >= cap) {
        if (overflow) atomicOr(overflow, 1);
        break;
      }
      colidx[pos++] = w * 64 + b;
      bits &= bits - 1;
    }
    base += __shfl(x, 63);
  }
}

// out = mask^T (n x n bits).  One wave per 64 x 64 bit tile: lane r loads word (64*tr + r, tc),
// bit b of every lane is gathered with a ballot into the word of output row 64*tc + b.
__global__ void mask_transpose_kernel(const unsigned long long* __restrict__ in, int n, int words,
                                      unsigned long long* __restrict__ out) {
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int ntile = words * words;
  if (tile >= ntile) return;  // whole wave exits together
  const int tr = tile / words, tc = tile - tr * words;
  const int lane = threadIdx.x & 63;
  const int r = tr * 64 + lane;
  const unsigned long long w = (r < n) ? in[(long)r * words + tc] : 0ull;
  unsigned long long mine = 0;
#pragma unroll
  for (int b = 0; b < 64; ++b) {
    const unsigned long long t = __ballot((w >> b) & 1ull);
    if (lane == b) mine = t;
  }
  const int orow = tc * 64 + lane;
  if (orow < n) out[(long)orow * words + tr] = mine;
}

// dense (n x n) <- bitmask; T = double (one modality, matrix_operations.py:135) or long long
// (fused, :138).  One thread per 4 columns... simple: one thread per element group of 64.
template <typename T>
__global__ void mask_to_dense_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                     T* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n * n;
  if (gid >= total) return;
  const int row = (int)(gid / n), col = (int)(gid - (long)row * n);
  out[gid] = (T)((mask[(long)row * words + (col >> 6)] >> (col & 63)) & 1ull);
}

// bitmask <- dense (nonzero -> 1).  One wave per (row, word): 64 columns per ballot.
template <typename T>
__global__ void dense_to_mask_kernel(const T* __restrict__ dense, int n, long ld, int words,
                                     unsigned long long* __restrict__ mask, int* __restrict__ nonbinary) {
  const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wid >= (long)n * words) return;
  const int row = (int)(wid / words), w = (int)(wid - (long)row * words);
  const int lane = threadIdx.x & 63;
  const int col = w * 64 + lane;
  bool nz = false;
  if (col < n) {
    const T v = dense[(long)row * ld + col];
    nz = (v != (T)0);
    if (nz && v != (T)1) atomicOr(nonbinary, 1);
  }
  const unsigned long long bal = __ballot(nz);
  if (lane == 0) mask[wid] = bal;
}

__global__ void zero_ints_kernel(int* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// Zero-fill by kernel, used inside captured sequences.  (Round 1 suspected captured hipMemsetAsync nodes of a memory
// fault; tools/repro_graph_memset.hip does not reproduce that on this stack -- the kernel stays because it is free.)
int zero_ints(int* p, long n, hipStream_t st) {
  hipLaunchKernelGGL(zero_ints_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int adj_csr_from_mask(const unsigned long long* mask, int n, int words, int* deg, int* rowptr, int* colidx,
                      int* stats, long cap, int* overflow, hipStream_t st) {
  hipLaunchKernelGGL(mask_degree_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, mask, n, words, deg);
  hipLaunchKernelGGL(degree_scan_kernel, dim3(1), dim3(1024), 0, st, deg, n, rowptr, stats, cap, overflow);
  hipLaunchKernelGGL(mask_to_csr_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, mask, n, words, rowptr, colidx, cap,
                     overflow);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int adj_transpose(const unsigned long long* mask, int n, int words, unsigned long long* out, hipStream_t st) {
  const long tiles = (long)words * words;
  hipLaunchKernelGGL(mask_transpose_kernel, dim3(cdiv(tiles, 4)), dim3(256), 0, st, mask, n, words, out);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // namespace mused

using namespace mused;

extern "C" {

// Replaces fuse_matrices (matrix_operations.py:134-141) on bitmasks: out = OR of M masks.
// `masks` is a HOST array of M device pointers (M <= 2 handled without a device pointer table).
int mused_adj_fuse(const unsigned long long* const* masks, int M, int n, int words, unsigned long long* out,
                   void* stream) {
  MUSED_REQUIRE(masks && M >= 1 && n > 0 && words >= (n + 63) / 64 && out, "mused_adj_fuse: bad arguments");
  const long total = (long)n * words;
  hipStream_t st = (hipStream_t)stream;
  if (M == 1) {
    if (out != masks[0]) MUSED_CHECK_HIP(hipMemcpyAsync(out, masks[0], total * 8, hipMemcpyDeviceToDevice, st));
    return MUSED_OK;
  }
  hipLaunchKernelGGL(mask_or2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, masks[0], masks[1], total, out);
  MUSED_LAUNCH_CHECK();
  for (int m = 2; m < M; ++m) {
    hipLaunchKernelGGL(mask_or2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, out, masks[m], total, out);
    MUSED_LAUNCH_CHECK();
  }
  return MUSED_OK;
}

// deg[i] = number of ones in row i; rowptr (n+1) = exclusive scan; stats[0] = max degree
// (= R of main.py:61 for a 0/1 matrix), stats[1] = nnz.  All outputs on the device.
int mused_adj_degrees(const unsigned long long* mask, int n, int words, int* deg, int* rowptr, int* stats,
                      void* stream) {
  MUSED_REQUIRE(mask && deg && rowptr && stats && n > 0 && words >= (n + 63) / 64, "mused_adj_degrees: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_degree_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, mask, n, words, deg);
  MUSED_LAUNCH_CHECK();
  hipLaunchKernelGGL(degree_scan_kernel, dim3(1), dim3(1024), 0, st, deg, n, rowptr, stats, 0l, (int*)nullptr);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// colidx <- neighbour lists (ascending) given rowptr from mused_adj_degrees.
int mused_adj_csr_fill(const unsigned long long* mask, int n, int words, const int* rowptr, int* colidx,
                       void* stream) {
  MUSED_REQUIRE(mask && rowptr && colidx && n > 0, "mused_adj_csr_fill: bad arguments");
  hipLaunchKernelGGL(mask_to_csr_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, mask, n, words,
                     rowptr, colidx, 0l, (int*)nullptr);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int mused_adj_transpose(const unsigned long long* mask, int n, int words, unsigned long long* out, void* stream) {
  MUSED_REQUIRE(mask && out && mask != out && n > 0 && words == (n + 63) / 64,
                "mused_adj_transpose: need distinct buffers and words == ceil(n/64)");
  const long tiles = (long)words * words;
  hipLaunchKernelGGL(mask_transpose_kernel, dim3(cdiv(tiles, 4)), dim3(256), 0, (hipStream_t)stream, mask, n,
                     words, out);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// Dense export for the NumPy-compatible call surface: dtype MUSED_F64 or MUSED_I64.
int mused_adj_to_dense(const unsigned long long* mask, int n, int words, int dtype, void* out, void* stream) {
  MUSED_REQUIRE(mask && out && n > 0, "mused_adj_to_dense: bad arguments");
  const long total = (long)n * n;
  if (dtype == MUSED_F64)
    hipLaunchKernelGGL(mask_to_dense_kernel<double>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       mask, n, words, (double*)out);
  else if (dtype == MUSED_I64)
    hipLaunchKernelGGL(mask_to_dense_kernel<long long>, dim3(cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, mask, n, words, (long long*)out);
  else {
    set_error("mused_adj_to_dense: unsupported dtype %d", dtype);
    return MUSED_ERR_UNSUPPORTED;
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// Dense import (nonzero -> edge).  *nonbinary (device int, zeroed by the caller) is set to 1
// if an entry other than 0/1 is met.
int mused_adj_from_dense(const void* dense, int dtype, int n, long ld, int words, unsigned long long* mask,
                         int* nonbinary, void* stream) {
  MUSED_REQUIRE(dense && mask && nonbinary && n > 0 && ld >= n && words >= (n + 63) / 64,
                "mused_adj_from_dense: bad arguments");
  const long waves = (long)n * words;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MUSED_F64)
    hipLaunchKernelGGL(dense_to_mask_kernel<double>, dim3(cdiv(waves, 4)), dim3(256), 0, st, (const double*)dense,
                       n, ld, words, mask, nonbinary);
  else if (dtype == MUSED_I64)
    hipLaunchKernelGGL(dense_to_mask_kernel<long long>, dim3(cdiv(waves, 4)), dim3(256), 0, st,
                       (const long long*)dense, n, ld, words, mask, nonbinary);
  else if (dtype == MUSED_F32)
    hipLaunchKernelGGL(dense_to_mask_kernel<float>, dim3(cdiv(waves, 4)), dim3(256), 0, st, (const float*)dense, n,
                       ld, words, mask, nonbinary);
  else {
    set_error("mused_adj_from_dense: unsupported dtype %d", dtype);
    return MUSED_ERR_UNSUPPORTED;
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // extern "C"
