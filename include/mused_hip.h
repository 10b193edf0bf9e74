/* libmused_hip -- C ABI of the MI355X (gfx950) hot path of mused.
 *
 * The reference (kelaendi/mused) has no FFI: its hot path sits behind two Python import lines,
 *   from matrix_operations import create_adjacency_matrix, fuse_matrices, perform_svd_reduction, ...   (main.py:5)
 *   from swfd import SeqBasedSWFD                                                                     (main.py:10)
 * This header is the drop-in boundary underneath those names: plain pointers and sizes, no torch
 * types.  Every pointer is a DEVICE pointer unless stated otherwise; `stream` is a hipStream_t;
 * outputs are caller-allocated; the library owns only the opaque handles.  Every function returns
 * 0 on success or a negative code (MUSED_ERR_*), the message is in mused_last_error().
 * No function synchronises with the host unless its comment says so.
 *
 * The Python host side that mirrors the reference's call surface on top of this ABI is
 * mused_amd/matrix_operations.py and mused_amd/swfd.py; INTEGRATION.md shows the binding.
 */
#ifndef MUSED_HIP_H
#define MUSED_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MUSED_OK 0
#define MUSED_ERR_ARG -1
#define MUSED_ERR_HIP -2
#define MUSED_ERR_STATE -3
#define MUSED_ERR_UNSUPPORTED -4

/* element types of caller-provided row data */
#define MUSED_F32 0
#define MUSED_F64 1
#define MUSED_I64 2
/* mused_swfd_append* only: rows of a 0/1 matrix given as BITMASK rows (bit c of a row = element c; row pitch in
 * 64-bit words) -- the fused adjacency of matrix_operations.py:134-141 fed to the sketch (main.py:65-67) without
 * the dense W x W matrix ever being built */
#define MUSED_BITS 3

/* metric of the pairwise score */
#define MUSED_METRIC_L2 0     /* squared Euclidean, matrix_operations.py:112-119 (sklearn NearestNeighbors) */
#define MUSED_METRIC_COSINE 1 /* negated cosine similarity, matrix_operations.py:106-108 */

const char* mused_last_error(void);
int mused_version(void);
/* async device-to-device copy on `stream` */
int mused_memcpy_d2d(void* dst, const void* src, long bytes, void* stream);

/* ---- a1 / a2: similarity -> k nearest rows ------------------------------------------------- */

/* out[i] = sum_j X[i][j]^2 (fp64).  Row norms of main.py:61 and of sklearn's Euclidean / cosine kernels. */
int mused_row_sq_norms(const void* X, int dtype, long n, int d, long ld, double* out, void* stream);

/* S (n x n fp64, pitch n): metric L2: max(0, |x_i|^2 - 2 x_i.x_j + |x_j|^2); COSINE: -(x_i.x_j)/(|x_i||x_j|)
 * (zero norms -> 1).  norms: n-double workspace.  fp64 MFMA GEMM with fused epilogue. */
int mused_pairwise_scores(const void* X, int dtype, long n, int d, long ld, int metric, double* norms, double* S,
                          void* stream);

/* Per row of S the k smallest entries, ties to the smaller column.  out_idx (n x k int32, ascending
 * columns) and out_mask (n x mask_words uint64 bitmask, the row's own column cleared --
 * matrix_operations.py:128) may each be NULL. */
int mused_select_k_smallest(const double* S, long ld, int n, int k, int* out_idx, unsigned long long* out_mask,
                            int mask_words, void* stream);

/* Replaces NearestNeighbors(n_neighbors=k).fit(X).kneighbors(X) + the adjacency write loop
 * (matrix_operations.py:118-130), and cosine_similarity + argsort[:, :k] (:106-108), for dense rows.
 * ws_scores: n*n doubles, ws_norms: n doubles. */
int mused_knn_topk(const void* X, int dtype, long n, int d, long ld, int k, int metric, double* ws_scores,
                   double* ws_norms, int* out_idx, unsigned long long* out_mask, int mask_words, void* stream);

/* The same selection WITHOUT the n x n score matrix: scores are filtered against a per-row running threshold as they
 * leave the MFMA tiles (symmetric tile grid walked by cyclic tile distance, candidate lists of <= cap entries per row,
 * exact (score, column) selection between phases).  ws: mused_knn_fused_ws_bytes(n, cap) bytes, cap <= 1024.
 * *overflow_out (device int, written on `stream`) != 0: a row collected more than cap candidates, outputs INVALID:
 * redo that window with mused_knn_topk. */
long mused_knn_fused_ws_bytes(long n, int cap);
int mused_knn_fused(const void* X, int dtype, long n, int d, long ld, int k, int metric, void* ws, long ws_bytes, int cap,
                    int* out_idx, unsigned long long* out_mask, int mask_words, int* overflow_out, void* stream);

/* ---- a1, metadata modality types (/root/reference/matrix_operations.py:22-89) -----------------------
 * Scores for mused_select_k_smallest (smaller = closer), all n x n fp64 with pitch n.
 * mused_record_scores: rec = n x 2 fp64 records.  kind MUSED_REC_LOCATION: (latitude, longitude) in degrees ->
 *   haversine distance in km, the arithmetic of haversine_distance (:250-263) operation by operation; replaces
 *   NearestNeighbors(metric=haversine_distance).kneighbors (:29-30).  kind MUSED_REC_TIME: (datetaken, dateupload) ->
 *   |d datetaken| + |d dateupload| (:40-50); replaces the per-row argsort loop (:39-54).
 * mused_jaccard_scores: tag sets as CSR (rowptr[n + 1], tags[]: ids < n_tags, unique per row) plus the transposed
 *   posting lists (postptr[n_tags + 1], postrow[]); S[i][j] = -|Ti & Tj| / |Ti | Tj| (0 if either set is empty,
 *   jaccard_similarity :245-248), S[i][i] = +1 (the reference scores a row against itself with -1 and sorts
 *   descending, :87-88).  n <= 65536.
 * mused_group_mask: ids[n] (< 0: no user name) -> adjacency bitmask of "same id, other row" (:56-71, 123-130). */
#define MUSED_REC_LOCATION 0
#define MUSED_REC_TIME 1
int mused_record_scores(const double* rec, int n, int kind, double* S, void* stream);
int mused_jaccard_scores(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int n_tags,
                         double* S, void* stream);
int mused_group_mask(const int* ids, int n, unsigned long long* out_mask, int mask_words, void* stream);
/* The same selections as mused_record_scores / mused_jaccard_scores followed by mused_select_k_smallest, WITHOUT the
 * n x n score matrix: the scores of a row are computed into LDS by the selection kernel itself (n <= 16384 records,
 * n <= 15000 tag sets).  Outputs as
 * mused_select_k_smallest (out_idx and out_mask may each be NULL; ties to the smaller row; own column cleared). */
int mused_record_knn(const double* rec, int n, int kind, int k, int* out_idx, unsigned long long* out_mask, int mask_words,
                     void* stream);
int mused_jaccard_knn(const int* rowptr, const int* tags, const int* postptr, const int* postrow, int n, int n_tags, int k,
                      int* out_idx, unsigned long long* out_mask, int mask_words, void* stream);

/* ---- a3 / a4: adjacency bitmasks -------------------------------------------------------------
 * An adjacency is n rows x words uint64 (words >= ceil(n/64)); bit j of row i <=> A[i][j] = 1. */

/* Replaces fuse_matrices (matrix_operations.py:134-141).  `masks`: HOST array of M device pointers. */
int mused_adj_fuse(const unsigned long long* const* masks, int M, int n, int words, unsigned long long* out,
                   void* stream);
/* deg[n], rowptr[n+1] (exclusive scan), stats = {max degree, nnz}; max degree = R of main.py:61. */
int mused_adj_degrees(const unsigned long long* mask, int n, int words, int* deg, int* rowptr, int* stats,
                      void* stream);
int mused_adj_csr_fill(const unsigned long long* mask, int n, int words, const int* rowptr, int* colidx,
                       void* stream);
int mused_adj_transpose(const unsigned long long* mask, int n, int words, unsigned long long* out, void* stream);
/* dense n x n export (MUSED_F64: one modality, matrix_operations.py:135; MUSED_I64: fused, :138) */
int mused_adj_to_dense(const unsigned long long* mask, int n, int words, int dtype, void* out, void* stream);
/* dense import (nonzero -> edge); *nonbinary (device int, pre-zeroed) set if an entry is not 0/1 */
int mused_adj_from_dense(const void* dense, int dtype, int n, long ld, int words, unsigned long long* mask,
                         int* nonbinary, void* stream);

/* ---- f3: hopping windows (step_window_ratio > 1, main.py:32) with reuse across consecutive windows --------------------
 * The same outputs as mused_knn_fused for the window whose row 0 is stream row lo_abs, computed from what the previous call
 * on the SAME workspace (the window n_new rows earlier) left behind: rows that stay keep their candidate lists minus the
 * columns that left; only the tiles involving one of the n_new entering rows (the LAST n_new rows of X) are evaluated:
 * 1 - (1 - n_new / n)^2 of the similarity work.  n_new = 0: from scratch (first window, or after a flag).  *flag_out
 * (device int) != 0: outputs INVALID (bit 0 a list overflowed, bit 1 a kept list no longer proves its row's k smallest) --
 * repeat with n_new = 0.  Bit-identical to mused_knn_fused on the same window. */
int mused_knn_fused_hop(const void* X, int dtype, long n, int d, long ld, int k, int metric, void* ws, long ws_bytes,
                        int cap, long lo_abs, int n_new, int* out_idx, unsigned long long* out_mask, int mask_words,
                        int* flag_out, void* stream);

/* ---- a8: randomized truncated SVD (perform_svd_reduction, matrix_operations.py:143-147) ------ */

int mused_rsvd_create(int n_max, int r_max, long nnz_cap, int sweeps, void** handle);
int mused_rsvd_destroy(void* handle);
/* buffer the fused adjacency bitmask (pitch ceil(n/64)) must be written to before mused_rsvd_reduce */
unsigned long long* mused_rsvd_mask_buffer(void* handle);
/* Q0 = np.random.RandomState(seed).normal(size=(n, r)) uploaded by the host (sklearn:utils/extmath.py:297) */
int mused_rsvd_set_q0(void* handle, const double* Q0, int n, int r, void* stream);
/* out_embed (n x n_comp) = X @ Vt.T, out_sigma (n_comp) = singular_values_, out_components (n x n_comp,
 * may be NULL) = Vt.T after svd_flip. */
int mused_rsvd_reduce(void* handle, int n, int n_comp, int r, int n_iter, double* out_embed, double* out_sigma,
                      double* out_components, void* stream);
/* device int[4] raised by mused_rsvd_reduce (flags[0] != 0: more than nnz_cap edges -> lists truncated, result
 * invalid but memory-safe; flags[2]: weak Cholesky pivot, see mused_rsvd_set_mode; flags[3] != 0: the r x r eigensolve
 * gave up (work-queue timeout), result invalid); copy it on the same stream behind the call for a sync-free check */
const int* mused_rsvd_flags(void* handle);
/* How the eigenstep builds its bases.  0 (default): Cholesky-QR (Gram + Cholesky + triangular solve) with the
 * Householder chain recorded behind a weak-pivot flag: self-contained.  1: Cholesky-QR only -- flags[2] != 0 (third int of
 * mused_rsvd_flags) after a call means the result is INVALID (numerically rank-deficient panel): repeat the call on a
 * handle in mode 2.  2: the reference's chain (LU-normalised power iterations, Householder QR), launched kernel by kernel
 * without graph capture. */
int mused_rsvd_set_mode(void* handle, int mode);

/* BLOCKING: flags_out[0] != 0 -> more than nnz_cap edges; stats_out = {max out-deg, nnz, max in-deg, nnz} (HOST) */
int mused_rsvd_status(void* handle, int* flags_out, int* stats_out, void* stream);

/* building blocks of the eigenstep, exported for unit tests */
int mused_spmm_binary(const int* rowptr, const int* colidx, int n, const double* Q, long ldq, int r, double* Y,
                      long ldy, void* stream);
/* ws_int: n + ceil(n/16) ints, ws_f64: 4 * (r + n) doubles */
int mused_lu_permute_l(double* Y, int n, int r, long ld, int* ws_int, double* ws_f64, void* stream);
/* ws_f64: r + ceil(n/512) * r + 2 n doubles; n >= r */
int mused_qr_economic(double* Y, int n, int r, long ldy, double* Q, long ldq, double* ws_f64, void* stream);
/* BLOCKING (creates/destroys its plan): eigen-decomposition of `batch` symmetric n x n matrices, n even */
int mused_syevj_batched(const double* G, int n, int batch, int sweeps, double* evals, double* V, void* stream);
int mused_gemm_f64(int a_kc, int b_kc, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   int M, int N, int K, double alpha, void* stream);
int mused_gemm_f64_batched(int a_kc, int b_kc, const double* A, long lda, long strideA, const double* B, long ldb,
                           long strideB, double* C, long ldc, long strideC, int M, int N, int K, int batch,
                           double alpha, void* stream);

/* ---- a10 / f2: the Lloyd iterations of perform_clustering (matrix_operations.py:149-153, sklearn KMeans) ------------
 * The k-means++ seeding stays on the host (scikit-learn's own routine on the same RandomState stream); E / M steps and
 * the stopping rule of sklearn's _kmeans_single_lloyd run here in fp64 with fixed-order sums.  X: n x d fp64 embedding
 * (pitch ld), mean: its d column means, centers: k x d seeds of the CENTRED rows (in/out), tol = mean(var(X, 0)) * 1e-4.
 * labels_out: n int32 (device); info_out (HOST, 4 ints) = {iterations, 1 strict / 2 tol / 0 max_iter, empty-cluster flag
 * (result invalid: use scikit-learn for that window), 0}.  BLOCKING.  k * d <= 8192. */
long mused_kmeans_ws_bytes(int n, int d, int k);
int mused_kmeans_lloyd(const double* X, long ld, int n, int d, int k, const double* mean, double* centers, double tol,
                       int max_iter, int* labels_out, int* info_out, void* ws, long ws_bytes, void* stream);

/* ---- a5-a7: SeqBasedSWFD (swfd submodule; call sites main.py:62,65-67,70) ---------------------- */

/* SeqBasedSWFD(N=, R=, d=, sketch_dim=) */
int mused_swfd_create(long N, double R, int d, int sketch_dim, int sweeps, void** handle);
/* `lanes` independent sketch sets advanced in lockstep by the same launches (windows of `lanes`
 * contiguous blocks of the stream): sketches, queries and outputs get a leading lane dimension */
int mused_swfd_create_lanes(long N, double R, int d, int sketch_dim, int sweeps, int lanes, void** handle);
int mused_swfd_lanes(void* handle);
/* n_rows rows for every lane; lane b reads rows + b * lane_stride (elements) */
int mused_swfd_append_lanes(void* handle, const void* rows, int dtype, long n_rows, long ld, long lane_stride,
                            void* stream);
int mused_swfd_destroy(void* handle);
int mused_swfd_levels(void* handle);
/* live timing of the rotation eigensolver: HIP events around every Jacobi sweep graph; read is BLOCKING and
 * returns summed ms, number of osj_round_kernel launches covered, bytes one launch streams (HOST outputs) */
int mused_swfd_profile(void* handle, int on);
int mused_swfd_profile_read(void* handle, double* total_ms, long* launches, double* bytes_per_launch);
/* the same for sketches whose rotations run a direct eigensolver (csrc/trd.hip: orders 2 l <= 256; csrc/trdx.hip: 320 .. 1024): ms of the solver-chain launches (trd_a .. trd_d), their
 * number, matrices they solved; *direct = 0 -> the rotations run the Jacobi, use mused_swfd_profile_read (HOST outputs) */
int mused_swfd_profile_read_direct(void* handle, double* total_ms, long* launches, double* matrices_solved, int* direct,
                                   double* tridiag_ms /* may be NULL: the share of total_ms spent in trd_a_kernel */);
/* .fit(row) for n_rows rows at once (any batching gives the same sketch) */
int mused_swfd_append(void* handle, const void* rows, int dtype, long n_rows, long ld, void* stream);
/* .get(): out_sketch (lanes x sketch_dim x d), out_sigma (lanes x sketch_dim, may be NULL),
 * out_info (lanes x 2 = {level, delta}, may be NULL) */
int mused_swfd_query(void* handle, double* out_sketch, double* out_sigma, double* out_info, void* stream);
int mused_swfd_counters(void* handle, long* rows_seen, int* pending); /* HOST outputs */
/* BLOCKING (synchronises `stream`): *status_out (HOST) != 0 -> an eigensolve of this sketch gave up (bit 0: timeout of the
 * persistent work-queue solver); everything it has returned since is invalid.  Sticky. */
int mused_swfd_status(void* handle, int* status_out, void* stream);
/* state exchange between ranks (one half = the L sketches of kind 0 MAIN / 1 AUX) */
long mused_swfd_half_bytes(void* handle);
int mused_swfd_export_half(void* handle, int kind, void* dst, void* stream);
int mused_swfd_import_half(void* handle, int kind, const void* src, void* stream);
int mused_swfd_begin_epoch(void* handle, long rows_seen, const void* main_half, void* stream);
/* BLOCKING unit primitive: one FD rotation of a (2 l x d) buffer in place */
int mused_fd_rotate(double* buf, int ell, int d, double* sigma_out, int sweeps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSED_HIP_H */
