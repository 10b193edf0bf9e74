"""CPU restatement of the reference's `matrix_operations.py` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `mused_amd/` may import this module;
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` use it, and there only as the checker / the timed CPU baseline.

Parity pin: every function below is checked against golden vectors produced
by importing the reference's own `matrix_operations.py` in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`,
`tests/test_oracle_golden.py`).

Each function cites the reference lines it restates (paths relative to
/root/reference) and, where the reference delegates to scikit-learn 1.7.2 /
SciPy 1.15.3, the library lines whose arithmetic it follows
(`sklearn:` = site-packages/sklearn).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

# --------------------------------------------------------------------------
# a1 / a2: similarity -> k nearest rows -> directed adjacency
# --------------------------------------------------------------------------


def _select_k_smallest_mask(score: np.ndarray, k: int) -> np.ndarray:
    """Boolean (n, n) mask of the k smallest entries of every row.

    Ties at the k-th value are broken towards the smaller column index: that is
    what sklearn's ArgKmin max-heap does when candidates arrive in ascending
    column order (a candidate replaces the heap top only if strictly smaller,
    sklearn:utils/_heap.pyx `heap_push`), and it is the rule the HIP kernel
    implements.
    """
    n = score.shape[1]
    k = min(k, n)
    thr = np.partition(score, k - 1, axis=1)[:, k - 1]
    less = score < thr[:, None]
    eq = score == thr[:, None]
    need = k - less.sum(axis=1)
    take_eq = eq & (np.cumsum(eq, axis=1) <= need[:, None])
    return less | take_eq


def sq_euclidean_scores(X: np.ndarray) -> np.ndarray:
    """||x_i||^2 - 2 x_i.x_j + ||x_j||^2, clamped at 0, in float64.

    sklearn:metrics/_pairwise_distances_reduction/_argkmin.pyx.tp
    (EuclideanArgKmin._compute_and_reduce_distances_on_chunks: squared norms
    + the -2 X Y^T GEMM "middle term", `max(0, ...)` clamp).
    """
    X = np.asarray(X, dtype=np.float64)
    n2 = np.einsum("ij,ij->i", X, X)
    D = X @ X.T
    D *= -2.0
    D += n2[:, None]
    D += n2[None, :]
    np.maximum(D, 0.0, out=D)
    return D


def cosine_scores(X: np.ndarray) -> np.ndarray:
    """-(cosine similarity); smaller = more similar.

    sklearn:metrics/pairwise.py `cosine_similarity` = `normalize(X)` (zero
    norms replaced by 1, sklearn:preprocessing/_data.py `normalize` /
    `_handle_zeros_in_scale`) followed by X_n @ X_n.T.
    Reference call site: matrix_operations.py:106-108 (`argsort(-text_sim)`).
    """
    X = np.asarray(X, dtype=np.float64)
    nrm = np.sqrt(np.einsum("ij,ij->i", X, X))
    nrm[nrm == 0.0] = 1.0
    Xn = X / nrm[:, None]
    S = Xn @ Xn.T
    np.negative(S, out=S)
    return S


def knn_mask(data: np.ndarray, k: int, metric: str = "l2") -> tuple[np.ndarray, np.ndarray]:
    """(valid_indices, boolean neighbour mask among the valid rows).

    metric "l2": matrix_operations.py:112-119 (rows with a non-finite entry are
    dropped, k = max(1, k_basis) nearest rows INCLUDING the row itself).
    metric "cosine": dense analogue of matrix_operations.py:91-108
    (k_basis + 1 most similar rows, self normally among them).
    """
    data = np.asarray(data, dtype=np.float64)
    valid = np.where(np.all(np.isfinite(data), axis=1))[0]
    Xv = data[valid]
    if len(Xv) == 0:
        return valid, np.zeros((0, 0), dtype=bool)
    if metric == "l2":
        kk = max(1, k)
        if kk > len(Xv):
            # sklearn:neighbors/_base.py kneighbors: n_neighbors <= n_samples_fit
            raise ValueError(
                f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {kk}, "
                f"n_samples_fit = {len(Xv)}, n_samples = {len(Xv)}"
            )
        score = sq_euclidean_scores(Xv)
    elif metric == "cosine":
        kk = min(k + 1, len(Xv))
        score = cosine_scores(Xv)
    else:
        raise ValueError(f"unknown metric {metric!r}")
    return valid, _select_k_smallest_mask(score, kk)


def text_vectors(data):
    """matrix_operations.py:95-105: rows with any non-empty column are valid; title and description are
    joined with blanks standing in for empty fields and vectorised by scikit-learn's TfidfVectorizer (the same
    library call as the reference; default parameters).  Returns (valid_indices, dense (n_valid, vocab) float64)."""
    from sklearn.feature_extraction.text import TfidfVectorizer

    data = np.asarray(data)
    valid = np.where(np.any(data != "", axis=1))[0]
    vd = data[valid]
    if len(vd) == 0:
        return valid, np.zeros((0, 0))
    text = np.where(vd[:, 0] != "", vd[:, 0], " ") + " " + np.where(vd[:, 1] != "", vd[:, 1], " ")
    if not np.any(text != " "):
        return valid, np.zeros((len(vd), 0))
    return valid, np.asarray(TfidfVectorizer().fit_transform(text).todense(), dtype=np.float64)


def haversine_scores(rec):
    """matrix_operations.py:250-263 for every pair of (latitude, longitude) rows, evaluated with the `math` functions the
    reference calls, operation by operation (pure-Python double loop: small cases only)."""
    from math import asin, cos, radians, sin, sqrt

    n = len(rec)
    S = np.zeros((n, n))
    rad = [(radians(float(a)), radians(float(b))) for a, b in rec]
    for i, (lat1, lon1) in enumerate(rad):
        for j, (lat2, lon2) in enumerate(rad):
            dlat = lat2 - lat1
            dlon = lon2 - lon1
            a = sin(dlat / 2) ** 2 + cos(lat1) * cos(lat2) * sin(dlon / 2) ** 2
            S[i, j] = 2 * asin(sqrt(a)) * 6371
    return S


def time_scores(rec):
    """matrix_operations.py:40-50: |datetaken_j - datetaken_i| + |dateupload_j - dateupload_i|."""
    rec = np.asarray(rec, dtype=np.float64)
    return np.abs(rec[None, :, 0] - rec[:, None, 0]) + np.abs(rec[None, :, 1] - rec[:, None, 1])


def jaccard_scores(tag_column):
    """matrix_operations.py:80-88 + jaccard_similarity (:245-248), negated so that smaller = more similar; a row
    scores +1 against itself (the reference gives it similarity -1 and sorts descending)."""
    sets = [set(t) if t else set() for t in tag_column]
    n = len(sets)
    S = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i == j:
                S[i, j] = 1.0
            elif sets[i] and sets[j]:
                S[i, j] = 0.0 - len(sets[i] & sets[j]) / len(sets[i] | sets[j])
    return S


def metadata_scores(data, modality_type, k_basis):
    """(valid row indices, score matrix among them (smaller = closer) or None, number of rows selected per row) of the
    metadata branches "location" (:22-31), "time" (:33-54), "tags" (:73-89)."""
    data = np.asarray(data)
    if modality_type == "location":
        valid = np.where(~np.isnan(data.astype(np.float64)).any(axis=1))[0]
        return valid, (haversine_scores(data[valid].astype(np.float64)) if len(valid) else None), k_basis + 1
    if modality_type == "time":
        valid = np.where(~((data[:, 0] == 0.0) | (data[:, 1] == 0.0)))[0]
        return valid, (time_scores(data[valid]) if len(valid) else None), 3 * k_basis + 1
    if modality_type == "tags":
        valid = np.where(data[:, 0] != "")[0]
        return valid, (jaccard_scores(data[valid, 0]) if len(valid) else None), k_basis
    raise ValueError(modality_type)


def create_adjacency_matrix(data, modality_type, k_basis=50):
    """matrix_operations.py:14-20,112-132 for the dense numeric `case _`
    (any modality_type the reference does not special-case), the `text`
    branch (:91-110: TF-IDF + cosine + top-(k+1)), a dense "cosine" type
    (the cosine kernel of that branch, :106-108, applied to already-vectorised
    rows) and the metadata branches "location", "time", "username", "tags" (:22-89).

    Returns the (n, n) float64 0/1 matrix: A[i, j] = 1 iff j is among the
    selected neighbours of i and j != i (directed; :123-130).
    """
    data = np.asarray(data)
    n = len(data)
    A = np.zeros((n, n))
    if modality_type == "text":
        valid, V = text_vectors(data)
        if len(valid) and V.shape[1]:
            mask = _select_k_smallest_mask(cosine_scores(V), min(k_basis + 1, len(valid)))
            np.fill_diagonal(mask, False)
            A[np.ix_(valid, valid)] = mask
        return A
    if modality_type == "username":  # :56-71: every other row of the same (non-empty) user name
        valid = np.where(data[:, 0] != "")[0]
        names = data[valid, 0]
        mask = names[:, None] == names[None, :]
        np.fill_diagonal(mask, False)
        A[np.ix_(valid, valid)] = mask
        return A
    if modality_type in ("location", "time", "tags"):
        # the k closest rows; where the reference's pick between EQUAL scores is undefined (unstable argsort :53,88,
        # ball-tree traversal :30) the smaller row index wins, as in _select_k_smallest_mask
        valid, S, kk = metadata_scores(data, modality_type, k_basis)
        if S is not None and kk > 0:
            mask = _select_k_smallest_mask(S, min(kk, len(valid)))
            np.fill_diagonal(mask, False)
            A[np.ix_(valid, valid)] = mask
        return A
    metric = "cosine" if modality_type == "cosine" else "l2"
    valid, mask = knn_mask(data, k_basis, metric)
    if len(valid):
        np.fill_diagonal(mask, False)
        A[np.ix_(valid, valid)] = mask
    return A


# --------------------------------------------------------------------------
# a3 / a4: fusion and R
# --------------------------------------------------------------------------


def fuse_matrices(matrices):
    """matrix_operations.py:134-141: copy for one modality (float64),
    logical OR cast to int64 for two or more."""
    fused = matrices[0].copy()
    for m in matrices[1:]:
        fused = np.logical_or(fused, m)
        fused = fused.astype(int)
    return fused


def max_row_sq_norm(fused) -> float:
    """main.py:61: max_i ||fused[i, :]||^2."""
    return float(np.max(np.linalg.norm(fused, axis=1) ** 2))


# --------------------------------------------------------------------------
# a8: randomized truncated SVD (the "eigenstep")
# --------------------------------------------------------------------------


def randomized_svd_reduce(matrix, reduced_dim, seed, n_iter=5, n_oversamples=10, info=None):
    """matrix_operations.py:143-147 = TruncatedSVD(n_components=min(reduced_dim,
    n_cols-1), random_state=seed).fit_transform(matrix), restated from
    sklearn:decomposition/_truncated_svd.py:225-274 and
    sklearn:utils/extmath.py `_randomized_range_finder` (:287-357),
    `_randomized_svd` (:531-601), `svd_flip` (:895-953).

    Returns (X_transformed (n, n_comp), singular_values (n_comp,), Vt (n_comp, n)).
    """
    A = np.asarray(matrix)
    n_comp = min(reduced_dim, A.shape[1] - 1)
    rs = np.random.RandomState(seed)
    n_random = n_comp + n_oversamples
    # n_samples == n_features for the fused adjacency -> no transpose
    transpose = A.shape[0] < A.shape[1]
    M = A.T if transpose else A
    Q = rs.normal(size=(M.shape[1], n_random))
    if np.issubdtype(M.dtype, np.floating):
        Q = Q.astype(M.dtype, copy=False)
    for _ in range(n_iter):  # n_iter=5 > 2 -> "LU" normaliser
        Q, _ = scipy.linalg.lu(M @ Q, permute_l=True, check_finite=False)
        Q, _ = scipy.linalg.lu(M.T @ Q, permute_l=True, check_finite=False)
    Q, _ = scipy.linalg.qr(M @ Q, mode="economic", check_finite=False)
    B = Q.T @ M
    Uhat, s, Vt = scipy.linalg.svd(B, full_matrices=False, lapack_driver="gesdd")
    if info is not None:  # test tooling: all n_comp + n_oversamples singular values of B (is the cut inside a multiple one?)
        info["sigma_all"] = np.array(s)
    U = Q @ Uhat
    if transpose:
        U, s, Vt = Vt[:n_comp, :].T, s[:n_comp], U[:, :n_comp].T
    else:
        U, s, Vt = U[:, :n_comp], s[:n_comp], Vt[:n_comp, :]
    # svd_flip(U, VT, u_based_decision=False)
    idx = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    Vt = Vt * signs[:, None]
    X_new = A @ Vt.T
    return X_new, s, Vt


def perform_svd_reduction(matrix, reduced_dim, seed):
    return randomized_svd_reduce(matrix, reduced_dim, seed)[0]


# --------------------------------------------------------------------------
# a10: consumers that turn the embedding into event indices (host side)
# --------------------------------------------------------------------------


def perform_clustering(matrix, n_clusters, seed):
    """matrix_operations.py:149-153.  The reference calls sklearn's KMeans;
    the consumer is kept as that same call."""
    from sklearn.cluster import KMeans

    return KMeans(n_clusters=n_clusters, random_state=seed).fit_predict(matrix)


def is_feasible(cost_matrix) -> bool:
    """matrix_operations.py:226-233."""
    inf = np.isinf(cost_matrix)
    if np.all(inf):
        return False
    if np.any(np.all(inf, axis=1)):
        return False
    if np.any(np.all(inf, axis=0)):
        return False
    return True


def match_clusters(prev_clusters, new_clusters, method="hungarian", min_overlap=5):
    """matrix_operations.py:155-185 + hungarian_matching :212-224
    (positional overlap counts -> -overlap / +inf costs -> LSA -> relabel)."""
    if prev_clusters is None or len(prev_clusters) == 0:
        return new_clusters
    prev_clusters = np.asarray(prev_clusters)
    new_clusters = np.asarray(new_clusters)
    up = np.unique(prev_clusters)
    un = np.unique(new_clusters)
    cost = np.full((len(up), len(un)), np.inf)
    for i, p in enumerate(up):
        for j, n in enumerate(un):
            ov = np.sum((prev_clusters == p) & (new_clusters == n))
            cost[i, j] = -ov if ov >= min_overlap else np.inf
    if not is_feasible(cost):
        return new_clusters
    if method != "hungarian":
        raise ValueError("Invalid method. Choose 'hungarian' or 'pot'.")
    r, c = linear_sum_assignment(cost)
    mapping = {un[cc]: up[rr] for rr, cc in zip(r, c)}
    return np.array([mapping.get(x, x) for x in new_clusters])


# --------------------------------------------------------------------------
# a9: the window loop (main.py:13-130), with the SWFD class injected
# --------------------------------------------------------------------------


def process_streaming_data(
    data_modalities,
    modality_types,
    window_size,
    reduced_dim,
    k_basis,
    seed,
    approach,
    true_labels,
    step_window_ratio=1,
    swfd_cls=None,
    trace=None,
):
    """Restates main.py:13-130 for approaches "sSVDMC"/"sSVDMC_hung" (SVD
    reduction) and "SWFDMC" (sketch reduction; `swfd_cls` supplies the
    SeqBasedSWFD implementation).  Returns the concatenated event labels
    (`all_clusters`, main.py:119,125).  `trace`, if a list, receives one dict
    per window (trigger index, singular values, labels before/after matching).
    """
    n = len(data_modalities[0])
    prev = None
    swfd = None
    out = []
    for i in range(n):
        # main.py:32 -- trigger
        if i + 1 >= window_size and (i + 1) * step_window_ratio % window_size == 0:
            lo = i + 1 - window_size
            tl = true_labels[lo : i + 1]
            n_clusters = len(np.unique(tl))  # main.py:41
            adjs = [
                create_adjacency_matrix(m[lo : i + 1], t, k_basis)
                for m, t in zip(data_modalities, modality_types)
            ]
            fused = fuse_matrices(adjs)
            sig = None
            if approach == "SWFDMC":
                if swfd is None:  # main.py:60-62
                    R = max_row_sq_norm(fused)
                    swfd = swfd_cls(N=window_size, R=R, d=fused.shape[1], sketch_dim=reduced_dim)
                for r in range(fused.shape[0]):  # main.py:65-67
                    swfd.fit(fused[r, :].reshape(1, -1))
                reduced, sig, _, _ = swfd.get()
                if reduced.shape[0] != window_size:  # main.py:73-76
                    reduced = reduced.T
            else:
                rinfo = {} if trace is not None else None
                reduced, sig, _ = randomized_svd_reduce(fused, reduced_dim, seed, info=rinfo)
            clusters = perform_clustering(reduced, n_clusters, seed)
            if trace is not None:  # (recorded before the matching, which may raise: main.py:331 lets that propagate)
                trace.append(dict(trigger=i, sigma=np.asarray(sig), raw=np.asarray(clusters), n_clusters=n_clusters,
                                  reduced=np.asarray(reduced)))
                if approach != "SWFDMC":
                    trace[-1]["sigma_all"] = rinfo.get("sigma_all")
            matched = match_clusters(prev, clusters, method="hungarian", min_overlap=3)
            if matched is None or len(matched) == 0:  # main.py:114-116
                matched = np.full(window_size, 0)
            if trace is not None:
                trace[-1]["matched"] = np.asarray(matched)
            prev = matched
            out.extend(matched)
    return np.array(out)
