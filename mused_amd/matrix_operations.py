"""MI355X-native stand-in for the reference's `matrix_operations` module.

Same eight names `main.py:5` imports, same argument meaning, return types and error behaviour:

    create_adjacency_matrix, fuse_matrices, match_clusters, perform_dbscan_clustering,
    perform_svd_reduction, perform_clustering, perform_dbscan_incr_clustering,
    perform_hdbscan_clustering

The numerical hot path (similarity -> kNN adjacency, fusion, randomized-SVD eigenstep) runs in
libmused_hip on the GPU through `mused_amd.engine.WindowEngine`; there is no CPU fallback -- without
the built extension or without a HIP device these functions raise.  The consumers that turn the
embedding into event indices (k-means, Hungarian matching) stay on the
host and call scikit-learn / SciPy exactly where the reference does, so label parity reduces to
embedding parity.

Two call styles:
  * NumPy in / NumPy out  -- drop-in for the reference (dense n x n matrices cross PCIe);
  * device objects        -- pass / receive `engine.Adjacency` and torch CUDA tensors
                             (`adjacency_on_device`, `fuse_matrices` on Adjacency objects,
                             `svd_reduce_on_device`) and nothing W x W ever visits the host.

modality_type "text" (matrix_operations.py:91-110) vectorises the ('title', 'description') strings on the
host with the same scikit-learn TfidfVectorizer call as the reference and runs the cosine / top-(k+1)
kernel on the device; already vectorised rows can use modality_type="cosine".
The metadata modality types of the SED2012 stream keep their string handling on the host and score /
select on the device (csrc/meta.hip): "location" (haversine kNN, :22-31), "time" (:33-54), "username" (:56-71),
"tags" (Jaccard, :73-89).  Where the reference's own choice between EQUAL scores is undefined (unstable argsort,
ball-tree traversal) the smaller row index wins here.
"""
from __future__ import annotations

import numpy as np

from . import engine as _eng

_METADATA_TYPES = ("location", "time", "username", "tags")


def _metric_for(modality_type) -> str:
    """Similarity kernel of a DENSE modality type (the metadata types have their own scores)."""
    if modality_type in _METADATA_TYPES:
        raise ValueError(f"modality_type={modality_type!r} is a metadata type, not a dense-row metric")
    return "cosine" if modality_type in ("cosine", "text") else "l2"


def edges_per_row(modality_type, k_basis):
    """Upper bound of the ones a row of create_adjacency_matrix can hold, None when there is none ("username")."""
    k = int(k_basis)
    if modality_type == "username":
        return None
    if modality_type == "time":
        return 3 * k + 1
    if modality_type == "tags":
        return max(k, 0)
    if modality_type in ("location", "cosine", "text"):
        return k + 1
    return max(k, 1)


def adjacency_on_device(data, modality_type="", k_basis=50, engine=None) -> _eng.Adjacency:
    """Device-resident result of create_adjacency_matrix (matrix_operations.py:14-132, `case _`).

    Rows with a non-finite entry are excluded from the kNN and get empty rows / columns
    (matrix_operations.py:114-115, 126-127)."""
    import torch

    if modality_type == "text":
        return _text_adjacency(data, k_basis, engine)
    if modality_type in _METADATA_TYPES:
        return _metadata_adjacency(data, modality_type, k_basis, engine)
    metric = _metric_for(modality_type)
    if isinstance(data, torch.Tensor):
        X = _eng.to_device_rows(data)
        finite = torch.isfinite(X).all(dim=1)
        all_valid = bool(finite.all().item())
        valid_idx = None if all_valid else torch.nonzero(finite).flatten()
    else:
        a = np.asarray(data)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        fin = np.all(np.isfinite(a), axis=1)
        all_valid = bool(fin.all())
        valid_np = None if all_valid else np.where(fin)[0]
        X = _eng.to_device_rows(a if all_valid else a[valid_np])
        valid_idx = None if all_valid else torch.from_numpy(valid_np).to(X.device)
        if not all_valid:
            finite = None
    n = len(data)
    eng = engine or _eng.default_engine(n)
    if all_valid:
        return eng.knn_adjacency(X, k_basis, metric)
    if isinstance(data, torch.Tensor):
        X = X[valid_idx].contiguous()
    if X.shape[0] == 0:
        w = _eng.words_for(n)
        return _eng.Adjacency(torch.zeros((n, w), dtype=torch.int64, device=eng.device), n)
    return _scatter_valid(eng.knn_adjacency(X, k_basis, metric), valid_idx, n)


def _scatter_valid(sub: _eng.Adjacency, valid_idx, n: int) -> _eng.Adjacency:
    """Adjacency among the valid rows -> window coordinates (rare path: dense round trip on the device)."""
    import torch

    dense = torch.zeros((n, n), dtype=torch.float64, device=sub.mask.device)
    dense[valid_idx.unsqueeze(1), valid_idx.unsqueeze(0)] = sub.to_dense(torch.float64)
    return _eng.Adjacency.from_dense(dense)


def _text_adjacency(data, k_basis, engine=None) -> _eng.Adjacency:
    """matrix_operations.py:91-110: rows with a non-empty title or description are valid; TF-IDF of
    "title description" on the host (same TfidfVectorizer call), then cosine similarity and the k_basis + 1 most
    similar rows per row on the device (`MUSED_METRIC_COSINE`)."""
    import torch
    from sklearn.feature_extraction.text import TfidfVectorizer

    data = np.asarray(data)
    n = len(data)
    eng = engine or _eng.default_engine(max(n, 1))
    empty = lambda: _eng.Adjacency(torch.zeros((n, _eng.words_for(n)), dtype=torch.int64, device=eng.device), n)
    valid = np.where(np.any(data != "", axis=1))[0]
    vd = data[valid]
    if len(vd) == 0:
        return empty()
    text = np.where(vd[:, 0] != "", vd[:, 0], " ") + " " + np.where(vd[:, 1] != "", vd[:, 1], " ")
    if not np.any(text != " "):
        return empty()
    V = np.asarray(TfidfVectorizer().fit_transform(text).todense(), dtype=np.float64)
    sub = eng.knn_adjacency(_eng.to_device_rows(V), k_basis, "cosine")
    if len(valid) == n:
        return sub
    return _scatter_valid(sub, torch.from_numpy(valid).to(eng.device), n)


def _metadata_adjacency(data, modality_type, k_basis, engine=None) -> _eng.Adjacency:
    """matrix_operations.py:22-89.  Row validity, k and the string handling follow the reference branch by branch on
    the host; scores and the selection run on the device."""
    import torch

    if isinstance(data, torch.Tensor):
        data = data.cpu().numpy()
    data = np.asarray(data)
    n = len(data)
    eng = engine or _eng.default_engine(max(n, 1))
    empty = lambda: _eng.Adjacency(torch.zeros((n, _eng.words_for(n)), dtype=torch.int64, device=eng.device), n)
    k = int(k_basis)
    if modality_type == "location":  # 'latitude', 'longitude'; k + 1 because a row is its own nearest neighbour
        valid = np.where(~np.isnan(data.astype(np.float64)).any(axis=1))[0]
        if len(valid) == 0:
            return empty()
        sub = eng.record_adjacency(data[valid], "location", min(k + 1, len(valid)))
    elif modality_type == "time":  # 'datetaken', 'dateupload'; 0.0 marks a missing stamp
        valid = np.where(~((data[:, 0] == 0.0) | (data[:, 1] == 0.0)))[0]
        if len(valid) == 0 or 3 * k + 1 <= 0:
            return empty()
        sub = eng.record_adjacency(data[valid], "time", min(3 * k + 1, len(valid)))
    elif modality_type == "username":
        valid = np.where(data[:, 0] != "")[0]
        if len(valid) == 0:
            return empty()
        _, ids = np.unique(data[valid, 0].astype(str), return_inverse=True)
        sub = eng.group_adjacency(ids)
    else:  # "tags"
        valid = np.where(data[:, 0] != "")[0]
        if len(valid) == 0 or k <= 0:
            return empty()
        vocab, rowptr, ids = {}, [0], []
        for tags in data[valid, 0]:
            tag_set = set(tags) if tags else set()
            ids.extend(sorted(vocab.setdefault(t, len(vocab)) for t in tag_set))
            rowptr.append(len(ids))
        sub = eng.jaccard_adjacency(rowptr, np.asarray(ids, dtype=np.int32), len(vocab), min(k, len(valid)))
    if len(valid) == n:
        return sub
    return _scatter_valid(sub, torch.from_numpy(valid).to(eng.device), n)


def create_adjacency_matrix(data, modality_type, k_basis=50):
    """matrix_operations.py:14: (n, n) float64 0/1 matrix, A[i, j] = 1 iff j is one of the
    k selected neighbours of i and j != i."""
    return adjacency_on_device(data, modality_type, k_basis).to_numpy()


def fuse_matrices(matrices):
    """matrix_operations.py:134-141.  A list of `Adjacency` objects gives a fused `Adjacency` (device
    OR of bitmasks); a list of ndarrays gives what the reference returns: a float64 copy for one
    matrix, the int64 logical OR for two or more."""
    if len(matrices) == 0:
        raise IndexError("list index out of range")  # matrices[0] in the reference
    if all(isinstance(m, _eng.Adjacency) for m in matrices):
        eng = _eng.default_engine(matrices[0].n)
        return eng.fuse(list(matrices))
    if len(matrices) == 1:
        return np.array(matrices[0], copy=True)  # plain .copy(): no arithmetic to accelerate
    adjs = [m if isinstance(m, _eng.Adjacency) else _eng.Adjacency.from_dense(np.asarray(m) != 0) for m in matrices]
    eng = _eng.default_engine(adjs[0].n)
    return eng.fuse(adjs).to_dense().cpu().numpy()


def max_row_sq_norm(fused) -> float:
    """R of main.py:61: max_i ||fused[i, :]||^2."""
    if isinstance(fused, _eng.Adjacency):
        return _eng.WindowEngine.max_row_sq_norm(fused)
    return _eng.WindowEngine.max_row_sq_norm(_eng.Adjacency.from_dense(fused))


def svd_reduce_on_device(matrix, reduced_dim, seed, nnz_cap=None, engine=None):
    """(embedding, singular_values) as fp64 CUDA tensors; `matrix` an Adjacency or a dense 0/1 matrix."""
    adj = matrix if isinstance(matrix, _eng.Adjacency) else _eng.Adjacency.from_dense(matrix)
    eng = engine or _eng.default_engine(adj.n)
    return eng.svd_reduce(adj, reduced_dim, seed, nnz_cap=nnz_cap)


def perform_svd_reduction(matrix, reduced_dim, seed):
    """matrix_operations.py:143-147: TruncatedSVD(n_components=min(reduced_dim, n_cols - 1),
    random_state=seed).fit_transform(matrix) for the 0/1 fused adjacency -> (n, n_comp) float64."""
    emb, _ = svd_reduce_on_device(matrix, reduced_dim, seed)
    return emb.cpu().numpy()


# ---- host-side consumers (keep on host, same library calls) ----------


def perform_clustering(matrix, n_clusters, seed):
    """matrix_operations.py:149-153."""
    from sklearn.cluster import KMeans

    if hasattr(matrix, "cpu"):
        matrix = matrix.cpu().numpy()
    return KMeans(n_clusters=n_clusters, random_state=seed).fit_predict(matrix)


_KM_WS = {}


def perform_clustering_on_device(emb_dev, n_clusters, seed, emb_host=None, stream=None):
    """The same labels as `perform_clustering` (matrix_operations.py:149-153) with the Lloyd iterations on the device
    (SURVEY 8 f2).  The k-means++ seeding is scikit-learn's own routine on the host, fed the same
    `RandomState(seed)` stream and the centred embedding exactly as `KMeans.fit` does (sklearn:cluster/_kmeans.py
    `fit`: tolerance from the raw rows, X -= X.mean(0), row norms, `_init_centroids` -> `_kmeans_plusplus`); assignment,
    centre update and the stopping rule of `_kmeans_single_lloyd` run in libmused_hip (csrc/kmeans.hip).
    emb_dev: (n, d) fp64 CUDA tensor; emb_host: its host copy if the caller already has one.  Returns int32 labels
    (NumPy).  Falls back to scikit-learn when a cluster runs empty (sklearn relocates it) or k * d > 8192."""
    import ctypes as C

    import torch
    from sklearn.cluster import kmeans_plusplus
    from sklearn.utils.extmath import row_norms

    from . import _lib

    n, d = emb_dev.shape
    X = np.ascontiguousarray(emb_host if emb_host is not None else emb_dev.cpu().numpy(), dtype=np.float64)
    if n_clusters * d > 8192 or n_clusters > n:
        return perform_clustering(X, n_clusters, seed)
    tol = float(np.mean(np.var(X, axis=0)) * 1e-4)   # KMeans._check_params_vs_input -> _tolerance, before centring
    mean = X.mean(axis=0)
    Xc = X - mean
    centers, _ = kmeans_plusplus(Xc, n_clusters, x_squared_norms=row_norms(Xc, squared=True),
                                 random_state=np.random.RandomState(seed))
    st = stream if stream is not None else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        key = (emb_dev.device, n, d, n_clusters, st.cuda_stream)
        ws = _KM_WS.get(key)
        if ws is None:
            ws = torch.empty(int(_lib.lib().mused_kmeans_ws_bytes(n, d, n_clusters)), dtype=torch.uint8, device=emb_dev.device)
            if len(_KM_WS) > 16:
                _KM_WS.clear()
            _KM_WS[key] = ws
        mean_d = torch.from_numpy(mean).to(emb_dev.device)
        cen_d = torch.from_numpy(np.ascontiguousarray(centers)).to(emb_dev.device)
        labels = torch.empty(n, dtype=torch.int32, device=emb_dev.device)
        info = (C.c_int * 4)()
        Xd = emb_dev if emb_dev.stride(1) == 1 else emb_dev.contiguous()
        _lib.call("mused_kmeans_lloyd", _eng.ptr(Xd), Xd.stride(0), n, d, n_clusters, _eng.ptr(mean_d), _eng.ptr(cen_d),
                  tol, 300, _eng.ptr(labels), info, _eng.ptr(ws), ws.numel(), C.c_void_p(st.cuda_stream))
        out = labels.cpu().numpy()
    if info[2]:
        return perform_clustering(X, n_clusters, seed)
    return out


def _overlap_costs(prev_clusters, new_clusters, min_overlap):
    up, un = np.unique(prev_clusters), np.unique(new_clusters)
    cost = np.full((len(up), len(un)), np.inf)
    for a, p in enumerate(up):
        sel = prev_clusters == p
        for b, q in enumerate(un):
            ov = int(np.count_nonzero(sel & (new_clusters == q)))
            if ov >= min_overlap:
                cost[a, b] = -ov
    return up, un, cost


def _feasible(cost) -> bool:
    inf = np.isinf(cost)
    return not (inf.all() or inf.all(axis=1).any() or inf.all(axis=0).any())


def match_clusters(prev_clusters, new_clusters, method="hungarian", min_overlap=5):
    """matrix_operations.py:155-185: relabel the new window's clusters by the previous window's
    labels through a minimum-cost assignment on positional overlap counts."""
    if prev_clusters is None or len(prev_clusters) == 0:
        return new_clusters
    prev_clusters = np.asarray(prev_clusters)
    new_arr = np.asarray(new_clusters)
    up, un, cost = _overlap_costs(prev_clusters, new_arr, min_overlap)
    if not _feasible(cost):
        return new_clusters
    if method == "hungarian":
        from scipy.optimize import linear_sum_assignment

        rows, cols = linear_sum_assignment(cost)
        relabel = {un[c]: up[r] for r, c in zip(rows, cols)}
        return np.array([relabel.get(c, c) for c in new_arr])
    if method == "pot":
        try:
            import ot
        except ImportError as e:  # the reference imports POT at module level (matrix_operations.py:12)
            raise ImportError("match_clusters(method='pot') needs the POT package") from e
        cost = np.abs(np.where(np.isinf(cost), 1e9, cost))
        cost = cost / cost.max()
        plan = ot.sinkhorn(np.full(len(up), 1.0 / len(up)), np.full(len(un), 1.0 / len(un)), cost, reg=0.1)
        rows, cols = np.where(plan > plan.max() * 0.5)
        relabel = {un[c]: up[r] for r, c in zip(rows, cols)}
        return np.array([relabel.get(c, c) for c in new_arr])
    raise ValueError("Invalid method. Choose 'hungarian' or 'pot'.")


def perform_dbscan_clustering(data, eps=0.5, min_samples=5):
    """matrix_operations.py:235-238."""
    from sklearn.cluster import DBSCAN

    return DBSCAN(eps=eps, min_samples=min_samples, metric="euclidean").fit_predict(data)


def perform_hdbscan_clustering(data, min_cluster_size=5, min_samples=2):
    """matrix_operations.py:240-243 (needs the optional `hdbscan` package, as the reference does)."""
    import hdbscan

    return hdbscan.HDBSCAN(min_cluster_size=min_cluster_size, min_samples=min_samples, metric="euclidean").fit_predict(data)


def perform_dbscan_incr_clustering(data, previous_centroids, previous_labels, eps=0.5, min_samples=5):
    """matrix_operations.py:265-298: DBSCAN on the window, clusters renamed after the closest
    centroid of the previous window."""
    from scipy.spatial.distance import cdist
    from sklearn.cluster import DBSCAN

    if not isinstance(data, np.ndarray):
        data = np.array(data, dtype=np.float32)
    if data.ndim != 2:
        return None, previous_centroids, previous_labels
    labels = DBSCAN(eps=eps, min_samples=min_samples, metric="euclidean").fit_predict(data)
    found = set(labels) - {-1}
    centroids = np.array([data[labels == c].mean(axis=0) for c in found])
    if previous_centroids is not None and len(previous_centroids) > 0:
        nearest = np.argmin(cdist(centroids, previous_centroids), axis=1)
        rename = {new: (previous_labels[old] if old < len(previous_labels) else -1) for new, old in enumerate(nearest)}
        labels = np.array([rename[l] if l in rename else l for l in labels])
    return labels, centroids, np.unique(labels)
