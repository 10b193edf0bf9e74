"""Pin the CPU oracle (oracle/mo_oracle.py) to golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import (COSINE_DENSE_CASES, METADATA_TIE_CASES, METADATA_TYPES, assert_valid_topk, cosine_dense_inputs,
                      load_golden, metadata_inputs, metadata_reference_adjacency, nbr_hash, regen_inputs, text_inputs)
from mused_amd import synth
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


@pytest.mark.slow
def test_reference_default_parameters_window_and_stream():
    """The reference's own operating point (/root/reference/main.py:305-313: window_size 2000, reduced_dim 50, k_basis 50):
    one window record and ten windows of its window loop (sSVDMC), oracle == reference."""
    from mused_amd import synth

    g = load_golden("refdef_blob_s0")
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    assert (W, ell, k) == (2000, 50, 50)
    _check_window(g, 0, mods, labels, W, ell, k, seed)
    g = load_golden("bench_refdef_blob_s0")
    n_windows, W, d, ell, k, seed = (int(x) for x in g["meta"])
    n_windows = 4   # (a prefix of the chain is a valid golden: stored per window)
    wins = [synth.stream_window("blob", t, W, d, seed) for t in range(n_windows)]
    assert [synth.array_digest(w[0]) for w in wins] == [str(x) for x in g["window_digest"][:n_windows]]
    X = np.concatenate([w[0] for w in wins]).astype(np.float64)
    labels = np.concatenate([w[1] for w in wins])
    out = mo.process_streaming_data([X], [""], W, ell, k, seed, "sSVDMC", labels)
    assert np.array_equal(np.asarray(out, dtype=np.int64).reshape(n_windows, W), g["labels"][:n_windows].astype(np.int64))


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


@pytest.mark.parametrize("tag", ["A", "B"])
@pytest.mark.parametrize("t", METADATA_TYPES)
def test_metadata_branches_match_reference(tag, t):
    """the reference's "location" / "time" / "username" / "tags" branches (matrix_operations.py:22-89) on
    synthetic SED2012-style columns.  Bit-equal adjacency wherever the reference's choice is defined; where it is decided
    by an unstable argsort between equal scores, the reference's adjacency must be A valid top-k of the oracle's scores
    (and so is the oracle's own, which breaks ties towards the smaller row)."""
    g = load_golden("metadata")
    cols, _, n, k = metadata_inputs(g, tag)
    A_ref = metadata_reference_adjacency(g, tag, t, n)
    A = mo.create_adjacency_matrix(cols[t], t, k)
    if (tag, t) not in METADATA_TIE_CASES:
        assert np.array_equal(A.astype(np.uint8), A_ref)
        assert nbr_hash(A) == str(g[f"{tag}_{t}_hash"])
    else:
        assert not np.array_equal(A.astype(np.uint8), A_ref)  # if this ever holds, move the case to the exact list
    if t != "username":
        valid, S, kk = mo.metadata_scores(cols[t], t, k)
        assert_valid_topk(A_ref, valid, S, kk)
        assert_valid_topk(A, valid, S, kk)


def test_metadata_degenerate_inputs():
    g = load_golden("metadata")
    assert np.array_equal(mo.create_adjacency_matrix(np.full((4, 2), np.nan), "location", 2), g["none_location"])
    assert np.array_equal(mo.create_adjacency_matrix(np.zeros((4, 2)), "time", 2), g["none_time"])
    assert np.array_equal(mo.create_adjacency_matrix(np.array([[""]] * 4), "username", 2), g["none_user"])
    few, _ = synth.metadata_stream(6, 2, missing=0.0)  # fewer valid rows than k
    for t in METADATA_TYPES:
        assert np.array_equal(mo.create_adjacency_matrix(few[t], t, 8).astype(np.uint8), g[f"few_{t}"])


def test_metadata_window_and_run_match_reference():
    """Fused (location OR username OR text) window through the eigenstep and k-means, and a whole run of the window
    loop over (location, username): hashes, sigma, labels as the reference produced them."""
    g = load_golden("metadata")
    n, k, ell, seed, sseed = (int(x) for x in g["win_meta"])
    cols, labels = synth.metadata_stream(n, sseed)
    text, _ = synth.text_stream(n, sseed)
    types_ = ["location", "username", "text"]
    adjs = [mo.create_adjacency_matrix(m, t, k) for m, t in zip([cols["location"], cols["username"], text], types_)]
    assert [nbr_hash(a) for a in adjs] == [str(h) for h in g["win_adj_hash"]]
    fused = mo.fuse_matrices(adjs)
    assert nbr_hash(fused) == str(g["win_fused_hash"]) and mo.max_row_sq_norm(fused) == float(g["win_R"])
    emb, sigma, _ = mo.randomized_svd_reduce(fused, ell, seed)
    np.testing.assert_allclose(sigma, g["win_sigma"], rtol=1e-10)
    assert np.array_equal(mo.perform_clustering(emb, len(np.unique(labels)), seed), g["win_labels"])
    n, W, ell, k, seed, sseed = (int(x) for x in g["run_meta"])
    cols, labels = synth.metadata_stream(n, sseed)
    out = mo.process_streaming_data([cols["location"], cols["username"]], ["location", "username"], W, ell, k, seed,
                                    "sSVDMC", labels)
    assert np.array_equal(out, g["run_clusters"])


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
