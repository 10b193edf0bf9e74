// Plain fp64 GEMM launchers on top of gemm_f64.h + library-wide error string.
#include <stdarg.h>

#include "gemm_f64.h"
#include "internal.h"

namespace mused {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_err; }

int gemm_f64_prepare_all() {
  int rc = 0;
#define PREP(a, b, v) rc |= gemm_f64_prepare_t<double, double, a, b, v, EpiStore>()
  PREP(true, true, true); PREP(true, true, false); PREP(true, false, true); PREP(true, false, false);
  PREP(false, true, true); PREP(false, true, false); PREP(false, false, true); PREP(false, false, false);
#undef PREP
  return rc ? MUSED_ERR_HIP : MUSED_OK;
}

int gemm_f64(bool a_kc, bool b_kc, const double* A, long lda, long strideA, const double* B, long ldb,
             long strideB, double* C, long ldc, long strideC, int M, int N, int K, int batch, double alpha,
             hipStream_t stream, const int* rep) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.strideA = strideA; g.strideB = strideB;
  g.M = M; g.N = N; g.K = K; g.splitk = 0; g.kchunk = 0;
  g.rep = rep;
  // Gram products (A A^T: same operand, same layout, square result): tiles on or above the diagonal + mirrored stores
  g.sym = (A == B && a_kc == b_kc && lda == ldb && strideA == strideB && M == N) ? 1 : 0;
  EpiStore epi{C, ldc, strideC, alpha};
  const bool vec = vec_ok<double>(A, lda, strideA) && vec_ok<double>(B, ldb, strideB);
  if (a_kc && b_kc) return gemm_f64_launch_t<double, double, true, true>(g, batch, epi, vec, stream);
  if (a_kc && !b_kc) return gemm_f64_launch_t<double, double, true, false>(g, batch, epi, vec, stream);
  if (!a_kc && b_kc) return gemm_f64_launch_t<double, double, false, true>(g, batch, epi, vec, stream);
  return gemm_f64_launch_t<double, double, false, false>(g, batch, epi, vec, stream);
}

int gemm_f64_splitk(bool a_kc, bool b_kc, const double* A, long lda, const double* B, long ldb, double* partial,
                    int M, int N, int K, int kchunk, int nsplit, hipStream_t stream) {
  MUSED_REQUIRE(kchunk % GEMM_BK == 0 && (long)kchunk * nsplit >= K, "gemm_f64_splitk: bad kchunk %d x %d for K=%d",
                kchunk, nsplit, K);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.lda = lda; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K; g.splitk = 1; g.kchunk = kchunk;
  EpiStore epi{partial, (long)N, (long)M * N, 1.0};
  const bool vec = vec_ok<double>(A, lda, 0) && vec_ok<double>(B, ldb, 0);
  if (a_kc && b_kc) return gemm_f64_launch_t<double, double, true, true>(g, nsplit, epi, vec, stream);
  if (a_kc && !b_kc) return gemm_f64_launch_t<double, double, true, false>(g, nsplit, epi, vec, stream);
  if (!a_kc && b_kc) return gemm_f64_launch_t<double, double, false, true>(g, nsplit, epi, vec, stream);
  return gemm_f64_launch_t<double, double, false, false>(g, nsplit, epi, vec, stream);
}

__global__ void splitk_reduce_kernel(const double* __restrict__ partial, int nsplit, long count,
                                     double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int z = 0; z < nsplit; ++z) s += partial[(long)z * count + i];  // fixed order: reproducible
  out[i] = s;
}

// Batched split-K: partial[(z * nsplit + sp)] (M x N each, ld = N) = opA(A[z]) opB(B[z]) over k in [sp * kchunk, (sp + 1) * kchunk);
// entries with rep[z] != z are skipped.  Sum with gemm_batched_splitk_reduce (fixed split order: results do not depend on the
// batch).  For products whose K loop is long and whose tiles do not fill the GPU (sketch Gram matrices at d = 10,000).
int gemm_f64_batched_splitk(bool a_kc, bool b_kc, const double* A, long lda, long strideA, const double* B, long ldb, long strideB,
                            double* partial, int M, int N, int K, int batch, int kchunk, int nsplit, hipStream_t stream,
                            const int* rep) {
  MUSED_REQUIRE(kchunk % GEMM_BK == 0 && nsplit >= 1 && (long)kchunk * nsplit >= K,
                "gemm_f64_batched_splitk: bad kchunk %d x %d for K=%d", kchunk, nsplit, K);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.strideA = strideA; g.strideB = strideB;
  g.M = M; g.N = N; g.K = K; g.splitk = 0; g.kchunk = kchunk; g.bsplit = nsplit;
  g.rep = rep;
  g.sym = (A == B && a_kc == b_kc && lda == ldb && strideA == strideB && M == N) ? 1 : 0;
  EpiStore epi{partial, (long)N, (long)M * N, 1.0};
  const bool vec = vec_ok<double>(A, lda, strideA) && vec_ok<double>(B, ldb, strideB);
  if (a_kc && b_kc) return gemm_f64_launch_t<double, double, true, true>(g, batch * nsplit, epi, vec, stream);
  if (a_kc && !b_kc) return gemm_f64_launch_t<double, double, true, false>(g, batch * nsplit, epi, vec, stream);
  if (!a_kc && b_kc) return gemm_f64_launch_t<double, double, false, true>(g, batch * nsplit, epi, vec, stream);
  return gemm_f64_launch_t<double, double, false, false>(g, batch * nsplit, epi, vec, stream);
}

__global__ void batched_splitk_reduce_kernel(const double* __restrict__ partial, int nsplit, long count, double* __restrict__ out,
                                             const int* __restrict__ rep) {
  const int z = blockIdx.y;
  if (rep && rep[z] != z) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double* p = partial + (long)z * nsplit * count + i;
  double s = 0.0;
  for (int sp = 0; sp < nsplit; ++sp) s += p[(long)sp * count];  // fixed order: reproducible
  out[(long)z * count + i] = s;
}

int gemm_batched_splitk_reduce(const double* partial, int nsplit, long count, double* out, int batch, const int* rep,
                               hipStream_t stream) {
  hipLaunchKernelGGL(batched_splitk_reduce_kernel, dim3(cdiv(count, 256), batch), dim3(256), 0, stream, partial, nsplit, count, out,
                     rep);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

int gemm_splitk_reduce(const double* partial, int nsplit, long count, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(count, 256)), dim3(256), 0, stream, partial, nsplit, count,
                     out);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // namespace mused

extern "C" {

const char* mused_last_error(void) { return mused::last_error(); }

int mused_version(void) { return 100; }

// Async device-to-device copy on `stream` (host code that only has raw pointers uses this).
int mused_memcpy_d2d(void* dst, const void* src, long bytes, void* stream) {
  MUSED_REQUIRE(dst && src && bytes >= 0, "mused_memcpy_d2d: bad arguments");
  MUSED_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MUSED_OK;
}

// Test/diagnostic entry: C = alpha * op(A) op(B), fp64.
int mused_gemm_f64(int a_kc, int b_kc, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   int M, int N, int K, double alpha, void* stream) {
  return mused::gemm_f64(a_kc != 0, b_kc != 0, A, lda, 0, B, ldb, 0, C, ldc, 0, M, N, K, 1, alpha,
                         (hipStream_t)stream);
}

int mused_gemm_f64_batched(int a_kc, int b_kc, const double* A, long lda, long strideA, const double* B, long ldb,
                           long strideB, double* C, long ldc, long strideC, int M, int N, int K, int batch,
                           double alpha, void* stream) {
  return mused::gemm_f64(a_kc != 0, b_kc != 0, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, M, N, K, batch,
                         alpha, (hipStream_t)stream);
}

}  // extern "C"
