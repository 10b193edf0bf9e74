// Host-side entry points shared between translation units of libmused_hip.
#pragma once
#include "common.h"

#include <mutex>

namespace mused {

// knn.hip: per-row selection of the k smallest scores computed on the fly (no n x n score matrix), see select_k_kernel
int select_from_records(const double* rec, int n, int kind, int k, int* out_idx, unsigned long long* out_mask,
                        int mask_words, hipStream_t stream);
int select_from_tag_sets(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int k,
                         int* out_idx, unsigned long long* out_mask, int mask_words, hipStream_t stream);
int select_max_fused_rows(bool tag_sets);

// One process-wide lock around every stream capture of this library AND around the calls that create or release HIP
// resources next to one (handle / plan creation and destruction: hipMalloc, hipFree, hipGraph(Exec)Destroy,
// hipStreamCreate / Destroy, hipFuncSetAttribute).  Round 2 locked only Begin..EndCapture; a handle re-created on one host
// thread then ran those calls beside another thread's capture (gpurun_out/r2_gputest23.log; tools/repro_capture_threads.hip
// asks the runtime which of them a ThreadLocal capture survives).  Recursive: creation paths nest (rsvd -> eig plan).
std::recursive_mutex& capture_mutex();
using CaptureLock = std::lock_guard<std::recursive_mutex>;


// C[z] = alpha * opA(A[z]) * opB(B[z]); fp64 in / fp64 out, MFMA f64 16x16x4.
//   a_kc: A stored [M][K] (true) or [K][M] (false);  b_kc: B stored [N][K] (true) or [K][N] (false).
int gemm_f64(bool a_kc, bool b_kc, const double* A, long lda, long strideA, const double* B, long ldb,
             long strideB, double* C, long ldc, long strideC, int M, int N, int K, int batch, double alpha,
             hipStream_t stream, const int* rep = nullptr);
// rep (device, batch ints, optional): entry z is computed only when rep[z] == z (duplicates are skipped)

// batched split-K (see gemm_f64.hip): partial results per (entry, split), then the fixed-order sum
int gemm_f64_batched_splitk(bool a_kc, bool b_kc, const double* A, long lda, long strideA, const double* B, long ldb, long strideB,
                            double* partial, int M, int N, int K, int batch, int kchunk, int nsplit, hipStream_t stream,
                            const int* rep = nullptr);
int gemm_batched_splitk_reduce(const double* partial, int nsplit, long count, double* out, int batch, const int* rep,
                               hipStream_t stream);
// Sets the dynamic-LDS attribute of every plain GEMM instantiation (call before stream capture).
int gemm_f64_prepare_all();

// Split-K variant for short-and-wide products (tiny M x N, long K): partial[z] for
// z < nsplit (each M x N, ld = N), then reduce with gemm_splitk_reduce.
int gemm_f64_splitk(bool a_kc, bool b_kc, const double* A, long lda, const double* B, long ldb, double* partial,
                    int M, int N, int K, int kchunk, int nsplit, hipStream_t stream);
int gemm_splitk_reduce(const double* partial, int nsplit, long count, double* out, hipStream_t stream);

// Y (n x r) <- P*L (first min(n,r) columns) ; pivstep: n + ceil(n/16) ints, prow: ws_f64_len doubles, at least
// r + ceil(n/16); with 4 (r + n) the blocked path runs (n <= 10240)
int lu_permute_l(double* Y, int n, int r, long ld, int* pivstep, double* prow, long ws_f64_len, hipStream_t st);
// Q (n x r) <- economic Householder Q of Y (destroyed); tau: r doubles, wpart: ceil(n/512)*r + 2 n doubles
// run_if (optional, device int): the whole chain is a no-op unless *run_if != 0 (fallback of the Cholesky-QR path)
int qr_economic(double* Y, int n, int r, long ldy, double* Q, long ldq, double* tau, double* wpart, hipStream_t st,
                const int* run_if = nullptr);
// Y = A Q for binary CSR A
int spmm_binary(const int* rowptr, const int* colidx, int n, const double* Q, long ldq, int r, double* Y, long ldy,
                hipStream_t stream);

// bitmask -> (deg, rowptr, colidx ascending); stats[0] = max degree, stats[1] = nnz; entries beyond
// `cap` (if cap > 0) are dropped and *overflow set.
int adj_csr_from_mask(const unsigned long long* mask, int n, int words, int* deg, int* rowptr, int* colidx,
                      int* stats, long cap, int* overflow, hipStream_t st);
int zero_ints(int* p, long n, hipStream_t st);
int adj_transpose(const unsigned long long* mask, int n, int words, unsigned long long* out, hipStream_t st);

// Batched symmetric eigensolver (cyclic Jacobi, one launch per rotation set).
struct EigPlan;
// own_graph: capture the sweep launches into a private hipGraph (set false when the caller
// captures a larger pipeline that contains this solve).
constexpr int EIG_PLAN_FIXED_SWEEPS = 1;  // always `sweeps` sweeps (no convergence flags)
constexpr int EIG_PLAN_NO_SORT = 2;       // no column sorting: column j of the result descends from column j of the input
constexpr int EIG_PLAN_TOP_HALF = 4;      // the caller reads only the n / 2 largest eigenpairs (FD rotation): orders the direct
                                          // solver covers (trd.hip) go to it, the Jacobi takes what its certificate rejects
constexpr int EIG_PLAN_TOP_NEED = 8;      // the caller reads the `need` largest eigenpairs, ALL of which must be certified (eigenstep)
constexpr int EIG_PLAN_TOP_FD = 16;       // as TOP_HALF with an explicit `need` (sketch query: the l largest of 3 l or 4 l)
int eig_plan_create(int n, int batch, int sweeps, bool own_graph, EigPlan** out, const int* rep = nullptr, int flags = 0,
                    int* err_out = nullptr, int need = 0);
// rep (device, batch ints, optional, read at every solve): matrix b is solved only when rep[b] == b
// err_out (device int, optional): OR-ed with 1 by a solve of the persistent work-queue solver that gave up waiting
// (timeout: the matrices are left partially rotated and the results of that solve are invalid)
void eig_plan_destroy(EigPlan* p);
// In: G (batch x n x n, symmetric) is copied into the plan's workspace.  Out: eigenvalues
// (unsorted, batch x n) and eigenvectors V (batch x n x n, column j <-> eigenvalue j).
int eig_plan_run(EigPlan* p, const double* G, double* evals, double* V, hipStream_t stream);
int eig_plan_profile(EigPlan* p, bool on);
int eig_plan_profile_read(EigPlan* p, double* total_ms, long* launches, double* bytes_per_launch);
// plans that run the direct solver: summed ms of its launches (HIP events around the kernel alone), number of launches,
// matrices it solved over all of them
int eig_plan_profile_read_direct(EigPlan* p, double* total_ms, long* launches, double* matrices_solved,
                                 double* tridiag_ms = nullptr);
bool eig_plan_direct_solver(EigPlan* p);  // the plan runs the direct solver (trd.hip) with the Jacobi as its fallback
double* eig_plan_input(EigPlan* p);  // (batch x n x n) device buffer the caller may fill directly
// Raw result of the one-sided solver (false: not available): cols[(b * ld + j) * ld + a] = lam_j u_j[a],
// lam[b * ld + j] = eigenvalue j; valid after eig_plan_run_inplace(p, nullptr, nullptr, ...).
bool eig_plan_columns(EigPlan* p, const double** cols, const double** lam, int* ld);
int eig_plan_run_inplace(EigPlan* p, double* evals, double* V, hipStream_t stream, bool allow_graph);

// trd.hip: direct solver for orders <= 256, the `need` <= 128 largest eigenpairs (tridiagonalisation + multisection +
// twisted factorisation + back-transformation); matrices column-major with leading dimension ldn (n <= ldn <= 256)
size_t trd_workspace_doubles(int batch);
int trd_prepare();
bool trd_supports(int n, int ldn, int need);
// the order-256 tridiagonalisation alone (the blocked solver's tail): see trd.hip
long trd_tail_ws_per();
long trd_tail_off_hs();
long trd_tail_off_tg();
int trd_tail_launch(const double* G, const int* rep, double* ws, int batch, hipStream_t st);
int trd_solve(double* Gc, int n, int ldn, int need, bool cert_all, int batch, const int* rep, int* done, double* ws,
              hipStream_t st, long long* dbg_clk = nullptr, unsigned long long* work = nullptr, hipEvent_t after_a = nullptr,
              double* lam_out = nullptr);  // lam_out (batch x ldn, optional): eigenvalues of the solved matrices (zeros behind them)

// trdx.hip: blocked direct solver for padded orders 320, 384, 448, 512 (need rounded up to 32 <= order / 2)
bool trdx_supports(int ldn, int need);
size_t trdx_workspace_doubles(int ldn, int batch);
int trdx_prepare(int ldn);
int trdx_solve(double* Gc, int ldn, int need, bool cert_all, int batch, const int* rep, int* done, int* act, int* jrep,
               int* nrej, double* ws, hipStream_t st, unsigned long long* work = nullptr, hipEvent_t after_a = nullptr,
               long long* prof = nullptr,   // prof (device, batch x 4, diagnostic): s_memtime ticks per phase of kernel A
               double* lam_out = nullptr);  // lam_out (batch x ldn, optional): eigenvalues of the solved matrices

}  // namespace mused
