// Y = A * Q for a 0/1 adjacency A given as CSR neighbour lists (ascending columns) and a
// dense row-major Q (n x r, fp64): Y[i, :] = sum_{j in nbr(i)} Q[j, :].
// This is the `A @ Q` / `A.T @ Q` of sklearn's randomized range finder
// (sklearn:utils/extmath.py:349-355) specialised to the binary fused adjacency that
// matrix_operations.py:143-147 is always called with.  Gather-bound: one wave per output
// row, lanes across the r columns (coalesced 8-B loads of whole Q rows, which stay
// L2 / Infinity-Cache resident: n*r*8 = 11 MB at n = 10^4, r = 138); neighbours are summed
// in list order, so results are bitwise reproducible.
#include "common.h"

namespace mused {

__global__ void spmm_binary_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx, int n,
                                   const double* __restrict__ Q, long ldq, int r, double* __restrict__ Y,
                                   long ldy) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const int beg = rowptr[row], end = rowptr[row + 1];
  for (int c0 = 0; c0 < r; c0 += 192) {
    const int ca = c0 + lane, cb = ca + 64, cc = ca + 128;
    const bool ha = ca < r, hb = cb < r, hc = cc < r;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int e = beg;
    for (; e + 2 <= end; e += 2) {
      const double* q0 = Q + (long)colidx[e] * ldq;
      const double* q1 = Q + (long)colidx[e + 1] * ldq;
      const double x0 = ha ? q0[ca] : 0.0, x1 = hb ? q0[cb] : 0.0, x2 = hc ? q0[cc] : 0.0;
      const double y0 = ha ? q1[ca] : 0.0, y1 = hb ? q1[cb] : 0.0, y2 = hc ? q1[cc] : 0.0;
      a0 += x0; a1 += x1; a2 += x2;
      a0 += y0; a1 += y1; a2 += y2;
    }
    if (e < end) {
      const double* q0 = Q + (long)colidx[e] * ldq;
      if (ha) a0 += q0[ca];
      if (hb) a1 += q0[cb];
      if (hc) a2 += q0[cc];
    }
    double* y = Y + (long)row * ldy;
    if (ha) y[ca] = a0;
    if (hb) y[cb] = a1;
    if (hc) y[cc] = a2;
  }
}

int spmm_binary(const int* rowptr, const int* colidx, int n, const double* Q, long ldq, int r, double* Y, long ldy,
                hipStream_t stream) {
  hipLaunchKernelGGL(spmm_binary_kernel, dim3(cdiv(n, 4)), dim3(256), 0, stream, rowptr, colidx, n, Q, ldq, r, Y,
                     ldy);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // namespace mused

extern "C" int mused_spmm_binary(const int* rowptr, const int* colidx, int n, const double* Q, long ldq, int r,
                                 double* Y, long ldy, void* stream) {
  MUSED_REQUIRE(rowptr && colidx && Q && Y && n > 0 && r > 0 && ldq >= r && ldy >= r, "mused_spmm_binary: bad arguments");
  return mused::spmm_binary(rowptr, colidx, n, Q, ldq, r, Y, ldy, (hipStream_t)stream);
}
