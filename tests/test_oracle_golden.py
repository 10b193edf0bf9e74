"""Pin the CPU oracle (oracle/mo_oracle.py) to golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import COSINE_DENSE_CASES, cosine_dense_inputs, load_golden, nbr_hash, regen_inputs, text_inputs
from oracle import mo_oracle as mo

WINDOW_CASES = ["c1_gauss_s0", "c1_gauss_s1", "c1_blob_s0", "c1_blob_s1", "c1_fd_s0", "c4s_twomod_s0"]


def _check_window(g, w, mods, labels, W, ell, k, seed, sigma_rtol=1e-9, emb_tol=1e-7):
    sl = slice(w * W, (w + 1) * W)
    adjs = [mo.create_adjacency_matrix(m[sl].astype(np.float64), "", k) for m in mods]
    for a, hh in zip(adjs, g[f"w{w}_adj_hash"]):
        assert nbr_hash(a) == str(hh)
    fused = mo.fuse_matrices(adjs)
    assert str(fused.dtype) == str(g[f"w{w}_fused_dtype"])
    assert nbr_hash(fused) == str(g[f"w{w}_fused_hash"])
    assert mo.max_row_sq_norm(fused) == pytest.approx(float(g[f"w{w}_R"]), rel=1e-12)
    emb, sigma, _ = mo.randomized_svd_reduce(fused, ell, seed)
    np.testing.assert_allclose(sigma, g[f"w{w}_sigma"], rtol=sigma_rtol)
    rows = g[f"w{w}_emb_rows"]
    scale = np.abs(g[f"w{w}_emb_sample"]).max()
    np.testing.assert_allclose(emb[rows], g[f"w{w}_emb_sample"], atol=emb_tol * scale)
    np.testing.assert_allclose(np.abs(emb).sum(axis=0), g[f"w{w}_emb_abs_colsum"], rtol=1e-7)
    km = mo.perform_clustering(emb, len(np.unique(labels[sl])), seed)
    assert np.array_equal(km.astype(np.int32), g[f"w{w}_kmeans_labels"])


@pytest.mark.parametrize("name", WINDOW_CASES)
def test_windows_match_reference(name):
    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    for w in range(n // W):
        _check_window(g, w, mods, labels, W, ell, k, seed)


@pytest.mark.slow
def test_mid_window_matches_reference():
    g = load_golden("c2m_blob_s0")
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    _check_window(g, 0, mods, labels, W, ell, k, seed)


@pytest.mark.parametrize(
    "name", ["c1_stream_blob_s0", "c1_stream_blob_s1", "c1_stream_gauss_s0", "c4s_stream_twomod_s0"]
)
def test_stream_event_labels_bit_exact(name):
    """Whole-run `all_clusters` of main.py:13-130 (sSVDMC) reproduced bit for bit."""
    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    out = mo.process_streaming_data(
        [m.astype(np.float64) for m in mods], [""] * len(mods), W, ell, k, seed, "sSVDMC", labels
    )
    assert out.dtype == np.int64 or out.dtype == np.int32
    assert np.array_equal(out.astype(np.int64), g["all_clusters"])
    assert hashlib.sha256(out.astype(np.int64).tobytes()).hexdigest() == str(g["labels_sha"])


@pytest.mark.parametrize("name,ratio", [("c1_stream_hop2_blob_s0", 2), ("c1_stream_hop4_gauss_s1", 4)])
def test_hopping_windows_match_reference(name, ratio):
    """step_window_ratio > 1 (main.py:32): overlapping windows, positional label matching -- `all_clusters` of the
    reference's own window loop reproduced bit for bit."""
    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    out = mo.process_streaming_data([m.astype(np.float64) for m in mods], [""], W, ell, k, seed, "sSVDMC", labels,
                                    step_window_ratio=ratio)
    assert np.array_equal(out.astype(np.int64), g["all_clusters"])


def test_text_branch_matches_reference():
    """a2: the reference's own `text` branch (matrix_operations.py:91-110) on synthetic string records."""
    g = load_golden("cosine")
    data, _, n, k = text_inputs(g)
    A = mo.create_adjacency_matrix(data, "text", k)
    assert nbr_hash(A) == str(g["text_adj_hash"])
    assert np.array_equal(A.sum(axis=1).astype(np.int32), g["text_deg"])
    invalid = np.where(~np.any(data != "", axis=1))[0]
    assert len(invalid) > 0 and A[invalid].sum() == 0 and A[:, invalid].sum() == 0
    assert np.array_equal(mo.create_adjacency_matrix(np.array([["", ""]] * 4), "text", 2).astype(np.uint8), g["blank_A"])


@pytest.mark.parametrize("tag,n,d,seed,k", COSINE_DENSE_CASES)
def test_cosine_kernel_matches_reference_arithmetic(tag, n, d, seed, k):
    """a2: sklearn `cosine_similarity` + `argsort(-sim)[:, :k+1]` + the write loop (matrix_operations.py:106-108,
    123-130) on dense rows == the oracle's "cosine" type."""
    g = load_golden("cosine")
    X = cosine_dense_inputs(g, tag, n, d, seed, k)
    A = mo.create_adjacency_matrix(X.astype(np.float64), "cosine", k)
    assert nbr_hash(A) == str(g[f"dense_{tag}_k{k}_hash"])


def test_edges():
    g = load_golden("edges")
    A = mo.create_adjacency_matrix(g["nonfinite_X"], "", 5)
    assert np.array_equal(A.astype(np.uint8), g["nonfinite_A"])
    assert A[3].sum() == 0 and A[:, 3].sum() == 0 and A[17].sum() == 0
    X = g["k1_X"]
    assert np.array_equal(mo.create_adjacency_matrix(X, "", 1).astype(np.uint8), g["k1_A"])
    assert np.array_equal(mo.create_adjacency_matrix(X, "", 0).astype(np.uint8), g["k0_A"])
    assert np.array_equal(mo.create_adjacency_matrix(X, "", 12).astype(np.uint8), g["kn_A"])
    with pytest.raises(ValueError):
        mo.create_adjacency_matrix(X, "", 13)
    A1 = g["fuse_A1"].astype(np.float64)
    A2 = g["fuse_A2"].astype(np.float64)
    F1 = mo.fuse_matrices([A1])
    F2 = mo.fuse_matrices([A1, A2])
    assert str(F1.dtype) == str(g["fuse1_dtype"]) and str(F2.dtype) == str(g["fuse2_dtype"])
    assert F1 is not A1 and np.array_equal(F1, A1)
    assert np.array_equal(F2.astype(np.uint8), g["fuse2"])
    Ad = mo.create_adjacency_matrix(g["demo_X"], "", 3)
    assert np.array_equal(Ad.astype(np.uint8), g["demo_A"])
    emb = mo.perform_svd_reduction(mo.fuse_matrices([Ad]), 2, 0)
    np.testing.assert_allclose(emb, g["demo_emb"], atol=1e-10)
    out = mo.match_clusters(g["match_prev"], g["match_new"], "hungarian", 3)
    assert np.array_equal(out, g["match_out"])
    out = mo.match_clusters(g["match_prev"], g["match_new_inf"], "hungarian", 3)
    assert np.array_equal(out, g["match_out_inf"])
    new = g["match_new"]
    assert mo.match_clusters(None, new) is new
    with pytest.raises(ValueError):
        mo.match_clusters(g["match_prev"], g["match_new"], "nope", 3)
