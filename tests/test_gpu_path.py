"""GPU parity of the hot path proper: kNN adjacency, fusion, eigenstep and SWFD against the CPU
oracle and the committed golden vectors (which come from the reference's own modules)."""
import numpy as np
import pytest

from conftest import (COSINE_DENSE_CASES, METADATA_TIE_CASES, METADATA_TYPES, assert_valid_topk, cosine_dense_inputs,
                      load_golden, metadata_inputs, metadata_reference_adjacency, nbr_hash, regen_inputs, text_inputs)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mused_amd.engine import WindowEngine

    e = WindowEngine(2048)
    yield e
    e.close()


def _bool_from_adj(adj):
    m = adj.mask.cpu().numpy().view(np.uint64)
    bits = np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")
    return bits[:, : adj.n].astype(bool)


# refdef_blob_s0: the reference's own default parameters (/root/reference/main.py:305-313: W = 2000, reduced_dim 50, k 50)
WINDOW_CASES = ["c1_gauss_s0", "c1_gauss_s1", "c1_blob_s0", "c1_blob_s1", "c1_fd_s0", "c4s_twomod_s0", "c2m_blob_s0",
                "refdef_blob_s0"]


@pytest.mark.parametrize("name", WINDOW_CASES)
def test_window_matches_reference_golden(eng, name):
    """adjacency hash, fused hash, R, singular values, embedding rows and k-means labels of the
    reference (golden) reproduced from the device path."""
    from mused_amd import matrix_operations as mo

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    for w in range(n // W):
        sl = slice(w * W, (w + 1) * W)
        adjs = [eng.knn_adjacency(torch.from_numpy(m[sl]).cuda(), k, "l2") for m in mods]
        for a, hh in zip(adjs, g[f"w{w}_adj_hash"]):
            assert nbr_hash(_bool_from_adj(a)) == str(hh)
        fused = eng.fuse(adjs)
        assert nbr_hash(_bool_from_adj(fused)) == str(g[f"w{w}_fused_hash"])
        assert str(fused.to_dense().cpu().numpy().dtype) == str(g[f"w{w}_fused_dtype"])
        assert eng.max_row_sq_norm(fused) == pytest.approx(float(g[f"w{w}_R"]), rel=1e-12)
        emb, sig = eng.svd_reduce(fused, ell, seed, nnz_cap=W * k * len(mods))
        flags, stats = eng.rsvd_status()
        assert flags == 0
        emb, sig = emb.cpu().numpy(), sig.cpu().numpy()
        np.testing.assert_allclose(sig, g[f"w{w}_sigma"], rtol=1e-9)  # north star: 1e-4 rel
        rows = g[f"w{w}_emb_rows"]
        scale = np.abs(g[f"w{w}_emb_sample"]).max()
        np.testing.assert_allclose(emb[rows], g[f"w{w}_emb_sample"], atol=1e-7 * scale)
        np.testing.assert_allclose(np.abs(emb).sum(axis=0), g[f"w{w}_emb_abs_colsum"], rtol=1e-7)
        km = mo.perform_clustering(emb, len(np.unique(labels[sl])), seed)
        assert np.array_equal(km.astype(np.int32), g[f"w{w}_kmeans_labels"])  # bit-exact event indices


def test_cosine_adjacency_matches_oracle(eng):
    from oracle import mo_oracle as omo

    rng = np.random.default_rng(4)
    X = rng.standard_normal((700, 96)).astype(np.float32)
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), 20, "cosine")
    ref = omo.create_adjacency_matrix(X, "cosine", 20)
    assert np.array_equal(_bool_from_adj(adj), ref.astype(bool))
    assert np.all(ref.sum(1) == 20)


@pytest.mark.parametrize("tag,n,d,seed,k", COSINE_DENSE_CASES)
def test_cosine_adjacency_matches_reference_golden(eng, tag, n, d, seed, k):
    """a2 pinned: the device cosine kernel against sklearn cosine_similarity + argsort (reference arithmetic,
    matrix_operations.py:106-108) recorded in tests/golden/cosine.npz."""
    g = load_golden("cosine")
    X = cosine_dense_inputs(g, tag, n, d, seed, k)
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), k, "cosine")
    assert nbr_hash(_bool_from_adj(adj)) == str(g[f"dense_{tag}_k{k}_hash"])


def test_text_modality_matches_reference_golden(eng):
    """The reference's `text` branch (host TF-IDF + device cosine top-(k+1)) on synthetic string records,
    incl. blank-title / blank-description / invalid rows."""
    from mused_amd import matrix_operations as mo

    g = load_golden("cosine")
    data, _, n, k = text_inputs(g)
    A = mo.create_adjacency_matrix(data, "text", k)
    assert A.dtype == np.float64 and A.shape == (n, n)
    assert nbr_hash(A) == str(g["text_adj_hash"])
    assert np.array_equal(mo.create_adjacency_matrix(np.array([["", ""]] * 4), "text", 2).astype(np.uint8), g["blank_A"])


@pytest.mark.parametrize("tag", ["A", "B"])
@pytest.mark.parametrize("t", METADATA_TYPES)
def test_metadata_branches_match_reference_and_oracle(tag, t):
    """SURVEY 8 f4 (matrix_operations.py:22-89): "location", "username" and the fractional-second "time" stream equal
    the reference's adjacency bit for bit; "tags" and whole-hour "time" (the reference's unstable argsort picks between
    equal scores there) equal the oracle (ties to the smaller row) and the reference's answer is a valid top-k of the
    same scores."""
    from mused_amd import matrix_operations as mo
    from oracle import mo_oracle as omo

    g = load_golden("metadata")
    cols, _, n, k = metadata_inputs(g, tag)
    A = mo.create_adjacency_matrix(cols[t], t, k)
    assert A.dtype == np.float64 and A.shape == (n, n)
    assert np.array_equal(A, omo.create_adjacency_matrix(cols[t], t, k))
    if (tag, t) not in METADATA_TIE_CASES:
        assert np.array_equal(A.astype(np.uint8), metadata_reference_adjacency(g, tag, t, n))
        assert nbr_hash(A) == str(g[f"{tag}_{t}_hash"])
    elif t != "username":
        valid, S, kk = omo.metadata_scores(cols[t], t, k)
        assert_valid_topk(A, valid, S, kk)


def test_metadata_degenerate_inputs_and_window_golden():
    from mused_amd import matrix_operations as mo
    from mused_amd import synth

    g = load_golden("metadata")
    assert np.array_equal(mo.create_adjacency_matrix(np.full((4, 2), np.nan), "location", 2), g["none_location"])
    assert np.array_equal(mo.create_adjacency_matrix(np.zeros((4, 2)), "time", 2), g["none_time"])
    assert np.array_equal(mo.create_adjacency_matrix(np.array([[""]] * 4), "username", 2), g["none_user"])
    few, _ = synth.metadata_stream(6, 2, missing=0.0)  # fewer valid rows than k
    for t in METADATA_TYPES:
        assert np.array_equal(mo.create_adjacency_matrix(few[t], t, 8).astype(np.uint8), g[f"few_{t}"])
    # fused (location OR username OR text) window: adjacency hashes, R, sigma, k-means labels of the reference
    n, k, ell, seed, sseed = (int(x) for x in g["win_meta"])
    cols, labels = synth.metadata_stream(n, sseed)
    text, _ = synth.text_stream(n, sseed)
    types_ = ["location", "username", "text"]
    adjs = [mo.adjacency_on_device(m, t, k) for m, t in zip([cols["location"], cols["username"], text], types_)]
    assert [nbr_hash(a.to_numpy()) for a in adjs] == [str(h) for h in g["win_adj_hash"]]
    fused = mo.fuse_matrices(adjs)
    assert nbr_hash(fused.to_numpy()) == str(g["win_fused_hash"]) and mo.max_row_sq_norm(fused) == float(g["win_R"])
    emb, sigma = mo.svd_reduce_on_device(fused, ell, seed)
    np.testing.assert_allclose(sigma.cpu().numpy(), g["win_sigma"], rtol=1e-8)
    km = mo.perform_clustering(emb.cpu().numpy(), len(np.unique(labels)), seed)
    assert np.array_equal(km, g["win_labels"])


def test_metadata_selection_needs_no_score_matrix():
    """mused_record_knn / mused_jaccard_knn (scores of a row computed into LDS by the selection kernel) == the score-matrix
    path (mused_record_scores / mused_jaccard_scores + mused_select_k_smallest), and allocate no n x n workspace."""
    from mused_amd import matrix_operations as mo
    from mused_amd import synth
    from mused_amd.engine import WindowEngine

    n, k = 1500, 40
    cols, _ = synth.metadata_stream(n, 5, events=6, users=30, integer_time=True)
    fused_eng, classic_eng = WindowEngine(n), WindowEngine(n)
    classic_eng.knn_mode = "classic"
    for t in ("location", "time", "tags"):
        a = mo.adjacency_on_device(cols[t], t, k, engine=fused_eng)
        b = mo.adjacency_on_device(cols[t], t, k, engine=classic_eng)
        assert torch.equal(a.mask, b.mask), t
    assert fused_eng._scores is None and classic_eng._scores is not None
    fused_eng.close()
    classic_eng.close()


def test_metadata_c_abi_index_outputs():
    """mused_record_knn / mused_jaccard_knn with BOTH outputs (neighbour indices and bitmask) against
    mused_record_scores / mused_jaccard_scores + mused_select_k_smallest, straight through the C ABI."""
    from mused_amd import _lib, synth
    from mused_amd.engine import ptr, stream_ptr, words_for

    n, k = 700, 23
    cols, _ = synth.metadata_stream(n, 9, missing=0.0, integer_time=True)
    w = words_for(n)
    S = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for kind, name in ((0, "location"), (1, "time")):
        rec = torch.from_numpy(np.ascontiguousarray(cols[name], dtype=np.float64)).cuda()
        i1 = torch.empty((n, k), dtype=torch.int32, device="cuda"); m1 = torch.empty((n, w), dtype=torch.int64, device="cuda")
        i2 = torch.empty_like(i1); m2 = torch.empty_like(m1)
        _lib.call("mused_record_knn", ptr(rec), n, kind, k, ptr(i1), ptr(m1), w, stream_ptr())
        _lib.call("mused_record_scores", ptr(rec), n, kind, ptr(S), stream_ptr())
        _lib.call("mused_select_k_smallest", ptr(S), n, n, k, ptr(i2), ptr(m2), w, stream_ptr())
        assert torch.equal(i1, i2) and torch.equal(m1, m2), name
    vocab, rowptr, ids = {}, [0], []
    for tags in cols["tags"][:, 0]:
        ids.extend(sorted(vocab.setdefault(t, len(vocab)) for t in set(tags)))
        rowptr.append(len(ids))
    rowptr = np.asarray(rowptr, dtype=np.int32); ids = np.asarray(ids, dtype=np.int32)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rowptr))
    order = np.argsort(ids, kind="stable")
    postptr = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=len(vocab)))]).astype(np.int32)
    dev = [torch.from_numpy(a).cuda() for a in (rowptr, ids, postptr, rows[order])]
    i1 = torch.empty((n, k), dtype=torch.int32, device="cuda"); m1 = torch.empty((n, w), dtype=torch.int64, device="cuda")
    i2 = torch.empty_like(i1); m2 = torch.empty_like(m1)
    _lib.call("mused_jaccard_knn", *(ptr(t) for t in dev), n, len(vocab), k, ptr(i1), ptr(m1), w, stream_ptr())
    _lib.call("mused_jaccard_scores", *(ptr(t) for t in dev), n, len(vocab), ptr(S), stream_ptr())
    _lib.call("mused_select_k_smallest", ptr(S), n, n, k, ptr(i2), ptr(m2), w, stream_ptr())
    assert torch.equal(i1, i2) and torch.equal(m1, m2)


def test_metadata_scores_at_window_size():
    """n = 2,500 rows (a quarter window; the oracle's haversine is a Python double loop): device adjacency == oracle for
    every metadata type, and the same-user relation has no degree bound (its CSR is sized from the real edge count)."""
    from mused_amd import matrix_operations as mo
    from mused_amd import synth
    from oracle import mo_oracle as omo

    n, k = 2500, 50
    cols, _ = synth.metadata_stream(n, 11, events=12, users=40)
    for t in METADATA_TYPES:
        A = mo.create_adjacency_matrix(cols[t], t, k)
        assert np.array_equal(A, omo.create_adjacency_matrix(cols[t], t, k)), t
    adj = mo.adjacency_on_device(cols["username"], "username", k)
    assert int(adj.degrees()[2][0].item()) > 3 * k + 1  # a busy user has more neighbours than any k-limited type
    emb, sigma = mo.svd_reduce_on_device(adj, 16, 0)  # nnz cap defaults to the real edge count
    assert np.isfinite(sigma.cpu().numpy()).all()


@pytest.mark.parametrize("name", WINDOW_CASES)
def test_device_kmeans_labels_match_reference_golden(eng, name):
    """f2: Lloyd iterations on the device with scikit-learn's own k-means++ seeding on the host == the reference's
    KMeans labels, bit for bit, on every golden window (structure-less Gaussian streams included: 14-40 iterations)."""
    from mused_amd import matrix_operations as mo

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    for w in range(n // W):
        sl = slice(w * W, (w + 1) * W)
        fused = eng.fuse([eng.knn_adjacency(torch.from_numpy(m[sl]).cuda(), k, "l2") for m in mods])
        emb, _ = eng.svd_reduce(fused, ell, seed, nnz_cap=W * k * len(mods))
        km = mo.perform_clustering_on_device(emb, len(np.unique(labels[sl])), seed)
        assert km.dtype == np.int32 and np.array_equal(km, g[f"w{w}_kmeans_labels"])


def test_device_kmeans_against_sklearn_on_hard_cases():
    """Unbalanced / overlapping / single-cluster inputs, k up to 40, d up to 200: device Lloyd == sklearn KMeans."""
    from sklearn.cluster import KMeans

    from mused_amd import matrix_operations as mo

    rng = np.random.default_rng(12)
    cases = [(3000, 16, 1), (3000, 16, 2), (2500, 64, 9), (4000, 200, 40), (700, 3, 5), (513, 128, 8)]
    for n, d, k in cases:
        X = rng.standard_normal((n, d)) + 3.0 * rng.standard_normal((k, d))[rng.integers(0, k, n)] * (rng.random((n, 1)) < 0.7)
        ref = KMeans(n_clusters=k, random_state=3).fit_predict(X)
        got = mo.perform_clustering_on_device(torch.from_numpy(X).cuda(), k, 3)
        assert np.array_equal(got, ref.astype(np.int32)), (n, d, k)


def test_rsvd_intermediate_components(eng):
    """Vt (after svd_flip) and the embedding against the oracle on a two-modality window."""
    from mused_amd import synth
    from oracle import mo_oracle as omo

    mods, _ = synth.two_modality_blob_stream(600, 24, 3, n_centres=5)
    adjs = [eng.knn_adjacency(torch.from_numpy(m).cuda(), 15, "l2") for m in mods]
    fused = eng.fuse(adjs)
    F = omo.fuse_matrices([omo.create_adjacency_matrix(m, "", 15) for m in mods])
    assert np.array_equal(_bool_from_adj(fused), F.astype(bool))
    emb, sig, comp = eng.svd_reduce(fused, 12, 7, want_components=True)
    e_ref, s_ref, vt_ref = omo.randomized_svd_reduce(F, 12, 7)
    np.testing.assert_allclose(sig.cpu().numpy(), s_ref, rtol=1e-10)
    np.testing.assert_allclose(comp.cpu().numpy(), vt_ref.T, atol=1e-8)
    np.testing.assert_allclose(emb.cpu().numpy(), e_ref, atol=1e-8 * np.abs(e_ref).max())


@pytest.mark.parametrize("ell", [134, 137, 138, 276])
def test_rsvd_two_block_cholesky_qr_sizes(ell):
    """r = reduced_dim + 10 in (143, 286] takes the two-block Cholesky-QR, whose projection scratch sits behind the packed
    factor: r = 144 .. 147 overran the workspace as round 3 sized it.  Singular values and embedding against the
    oracle's LU / Householder chain at the boundary sizes r = 144, 147, 148, 286."""
    from mused_amd import synth
    from mused_amd.engine import WindowEngine
    from oracle import mo_oracle as omo

    n, k = 900, 30
    X, _ = synth.blob_stream(n, 48, 5, n_centres=6)
    e = WindowEngine(n)
    try:
        adj = e.knn_adjacency(torch.from_numpy(X).cuda(), k, "l2")
        fused = e.fuse([adj])
        emb, sig = e.svd_reduce(fused, ell, 3, nnz_cap=n * k)
        flags, _ = e.rsvd_status()
        assert flags == 0
        F = omo.fuse_matrices([omo.create_adjacency_matrix(X.astype(np.float64), "", k)])
        e_ref, s_ref, _ = omo.randomized_svd_reduce(F, ell, 3)
        np.testing.assert_allclose(sig.cpu().numpy(), s_ref, rtol=1e-8)
        big = s_ref > 1e-6 * s_ref[0]
        np.testing.assert_allclose(emb.cpu().numpy()[:, big], e_ref[:, big], atol=1e-7 * np.abs(e_ref).max())
    finally:
        e.close()


def test_rsvd_tiny_window_more_random_columns_than_rows(eng):
    """main.py:318-324 demo sizes: n_components + 10 > n."""
    g = load_golden("edges")
    from mused_amd import matrix_operations as mo

    A = mo.create_adjacency_matrix(g["demo_X"], "", 3)
    assert np.array_equal(A.astype(np.uint8), g["demo_A"])
    emb = mo.perform_svd_reduction(mo.fuse_matrices([A]), 2, 0)
    np.testing.assert_allclose(emb, g["demo_emb"], atol=1e-9)


# ---------------------------------------------------------------- SWFD -------------------
def _swfd_pair(N, R, d, ell):
    from mused_amd.swfd import SeqBasedSWFD as Dev
    from oracle.swfd_oracle import SeqBasedSWFD as Ora

    return Dev(N=N, R=R, d=d, sketch_dim=ell), Ora(N=N, R=R, d=d, sketch_dim=ell)


def _compare(dev, ora, tag):
    Bd, sd, ld, dd = dev.get()
    Bo, so, lo, do = ora.get()
    assert ld == lo, f"{tag}: level {ld} vs {lo}"
    s0 = max(so[0], 1e-300)
    np.testing.assert_allclose(sd, so, rtol=0, atol=1e-8 * s0, err_msg=tag)  # north star: 1e-4 rel
    big = so > 1e-3 * s0
    np.testing.assert_allclose(sd[big], so[big], rtol=1e-6, err_msg=tag)
    np.testing.assert_allclose(Bd.T @ Bd, Bo.T @ Bo, rtol=0, atol=1e-8 * s0 * s0, err_msg=tag)
    assert abs(dd - do) <= 1e-8 * s0 * s0


@pytest.mark.parametrize("kind", ["gauss", "blob", "fd"])
def test_swfd_matches_oracle_over_a_stream(kind):
    from mused_amd import synth

    N, ell, d = 400, 8, 40
    X, _ = synth.make_stream(kind, 3 * N + 37, d, 3)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    dev, ora = _swfd_pair(N, R, d, ell)
    assert dev.L == ora.L
    t = 0
    for step in [1, 7, 120, 272, 1, 399, 400, 37]:  # ragged batching incl. epoch boundaries
        blk = X[t : t + step]
        dev.fit(blk)
        ora.fit(blk)
        t += step
        _compare(dev, ora, f"{kind} t={t}")
    dev.close()


def test_swfd_orders_padded_into_the_blocked_solver():
    """l = 150: rotations of order 300 (zero padded to 320) and queries of order 450 / 600 (padded to 512 / 640) -- orders that
    are not the blocked direct solver's own, solved embedded in the next one it has, its trailing 256 x 256 on the register-resident
    kernel.  Device == specification over an epoch end (parity unpinned, as for every SWFD test)."""
    from mused_amd import synth

    N, ell, d = 900, 150, 320
    X, _ = synth.make_stream("blob", N + 520, d, 4)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    dev, ora = _swfd_pair(N, R, d, ell)
    t = 0
    for step in [300, 450, 170, 500]:
        blk = X[t : t + step]
        dev.fit(blk)
        ora.fit(blk)
        t += step
        _compare(dev, ora, f"l=150 t={t}")
    dev.close()


def test_swfd_device_rows_int64_and_per_row_fit():
    """mused wiring: d = window size, rows of the int64 fused adjacency, one fit() per row (main.py:65-67)."""
    W, ell = 96, 6
    rng = np.random.default_rng(2)
    fused = (rng.random((2 * W, W)) < 0.08).astype(np.int64)
    R = float(np.max(np.linalg.norm(fused[:W], axis=1) ** 2))
    dev, ora = _swfd_pair(W, R, W, ell)
    for r in range(W):
        row = fused[r, :].reshape(1, -1)
        dev.fit(row)
        ora.fit(row)
    _compare(dev, ora, "per-row window 1")
    dev.fit(torch.from_numpy(fused[W:]).cuda())  # device-resident int64 block
    ora.fit(fused[W:])
    _compare(dev, ora, "device block window 2")
    red = dev.get()[0]
    assert red.shape == (ell, W) and red.T.shape == (W, ell)  # main.py:73-76 transposes this
    dev.close()


def test_swfd_expiry_and_dumps():
    """A heavy early direction forces dumps on the low levels and must be forgotten N rows later."""
    N, ell, d = 300, 6, 24
    rng = np.random.default_rng(0)
    heavy = np.zeros((N, d))
    heavy[:, 0] = 30.0 * (1 + 0.1 * rng.standard_normal(N))
    X = np.vstack([heavy, rng.standard_normal((2 * N + 11, d))])
    R = float((X**2).sum(1).max())
    dev, ora = _swfd_pair(N, R, d, ell)
    for t0 in range(0, len(X), 100):
        dev.fit(X[t0 : t0 + 100])
        ora.fit(X[t0 : t0 + 100])
        _compare(dev, ora, f"t={t0 + 100}")
    B = dev.get()[0]
    assert (B[:, 0] ** 2).sum() < 5 * N
    dev.close()


def test_swfd_state_exchange_between_ranks():
    """Windows sharded over two sketch objects (ranks): rank 1 starts window 1 from the AUX half of
    rank 0 and must equal the single sequential sketch."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD as Dev

    N, ell, d = 256, 8, 32
    X, _ = synth.blob_stream(2 * N, d, 5, n_centres=3)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    seq = Dev(N=N, R=R, d=d, sketch_dim=ell)
    seq.fit(X)
    B_seq, s_seq, l_seq, _ = seq.get()
    r0 = Dev(N=N, R=R, d=d, sketch_dim=ell)
    r0.fit(X[:N])
    blob = r0.export_half(1)
    r1 = Dev(N=N, R=R, d=d, sketch_dim=ell)
    r1.begin_epoch(N, blob)
    r1.fit(X[N:])
    B1, s1, l1, _ = r1.get()
    assert l1 == l_seq
    assert np.array_equal(s1, s_seq) and np.array_equal(B1, B_seq)
    for o in (seq, r0, r1):
        o.close()


def test_swfd_lanes_equal_independent_sketches():
    """B lanes advanced in lockstep by shared launches == B separate sketch objects."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD as Dev

    N, ell, d, B = 300, 8, 48, 3
    Xs = [synth.make_stream(kind, 2 * N + 50, d, 7 + i)[0] for i, kind in enumerate(["gauss", "blob", "fd"])]
    R = max(float((x.astype(np.float64) ** 2).sum(1).max()) for x in Xs)
    lanes = Dev(N=N, R=R, d=d, sketch_dim=ell, lanes=B)
    singles = [Dev(N=N, R=R, d=d, sketch_dim=ell) for _ in range(B)]
    X = torch.from_numpy(np.stack(Xs)).cuda()  # (B, n, d) float32
    t = 0
    for step in [130, 170, 1, 299, 50]:
        lanes.fit_lanes(X[:, t : t + step])
        for b in range(B):
            singles[b].fit(X[b, t : t + step])
        t += step
        Bl, sl, ll, dl = lanes.get()
        for b in range(B):
            Bs, ss, ls, ds = singles[b].get()
            assert int(ll[b]) == ls
            assert np.array_equal(Bl[b], Bs) and np.array_equal(sl[b], ss) and dl[b] == ds
    lanes.close()
    for sk in singles:
        sk.close()


# ---------------------------------------------------------------- BASELINE config-2 full size ----
@pytest.mark.parametrize("name", ["c2_blob_s0", "c2_gauss_s0", "c3_blob_s0", "c4_twomod_s0"])
def test_full_size_window_matches_reference_golden(name):
    """One full-size window of BASELINE.json configs[1] (W = 10,000, d = 1024, l = 128), configs[2] (d = 4096,
    l = 256) and configs[3] (two 512-d modalities, OR-fused, l = 128), k = 50: neighbour hashes, fused hash, R, the
    l singular values, embedding samples and the 10,000 k-means labels of the REFERENCE reproduced."""
    from mused_amd import matrix_operations as mo
    from mused_amd.engine import WindowEngine

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    eng = WindowEngine(W)
    adjs = []
    for m, hh in zip(mods, g["w0_adj_hash"]):
        adj = eng.knn_adjacency(torch.from_numpy(m).cuda(), k, "l2")
        assert nbr_hash(_bool_from_adj(adj)) == str(hh)
        adjs.append(adj)
    fused = eng.fuse(adjs)
    assert nbr_hash(_bool_from_adj(fused)) == str(g["w0_fused_hash"])
    assert eng.max_row_sq_norm(fused) == pytest.approx(float(g["w0_R"]), rel=1e-12)
    emb, sig, flags = eng.svd_reduce(fused, ell, seed, nnz_cap=W * k * len(mods), want_flags=True)
    assert int(flags.cpu()[0]) == 0
    emb, sig = emb.cpu().numpy(), sig.cpu().numpy()
    np.testing.assert_allclose(sig, g["w0_sigma"], rtol=1e-8)  # north star: 1e-4 rel
    rows = g["w0_emb_rows"]
    np.testing.assert_allclose(emb[rows], g["w0_emb_sample"], atol=1e-7 * np.abs(g["w0_emb_sample"]).max())
    km = mo.perform_clustering(emb, len(np.unique(labels)), seed)
    assert np.array_equal(km.astype(np.int32), g["w0_kmeans_labels"])  # bit-exact event indices
    kd = mo.perform_clustering_on_device(torch.from_numpy(emb).cuda(), len(np.unique(labels)), seed)
    assert np.array_equal(kd, g["w0_kmeans_labels"])                   # ... also with the Lloyd iterations on the device
    eng.close()


def test_headline_shape_swfd_lanes_match_oracle_and_single_sketches():
    """The configuration bench.py times -- N = 10,000, d = 1024, l = 128, several lanes advanced in lock-step
    with duplicate-level skipping on (the default) -- against oracle/swfd_oracle.py (sigma, level, final shrink, a sampled block
    of the covariance: tests/golden/swfd_headline_lanes.npz, written by make_swfd_fixtures.py from the specification -- parity
    unpinned) and, bit for bit, against single-lane handles, over the first 1,152 rows (9 rotations of all 28 sketches per
    lane) of three different windows of the benchmark stream."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD as Dev

    g = load_golden("swfd_headline_lanes")
    W, d, ell, seed, B = (int(x) for x in g["meta"][:5])
    steps = [int(x) for x in g["meta"][5:]]
    rows = sum(steps)
    Xs = [synth.stream_window("blob", t, W, d, seed)[0][:rows] for t in range(B)]
    assert [synth.array_digest(x) for x in Xs] == [str(x) for x in g["input_digest"]]
    R = float(g["R"])
    lanes = Dev(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
    singles = [Dev(N=W, R=R, d=d, sketch_dim=ell) for _ in range(B)]
    assert lanes.L == int(g["levels"]) == 14
    X = torch.from_numpy(np.stack(Xs)).cuda()
    idx = g["gram_idx"]
    t = 0
    for i, step in enumerate(steps):  # 5 whole blocks, then 4
        lanes.fit_lanes(X[:, t : t + step])
        for b in range(B):
            singles[b].fit(X[b, t : t + step])
        t += step
        Bl, sl, ll, dl = lanes.get()
        for b in range(B):
            Bs, ss, ls, ds = singles[b].get()
            assert int(ll[b]) == ls and np.array_equal(Bl[b], Bs) and np.array_equal(sl[b], ss) and dl[b] == ds
            so = g["sigma"][b][i]
            assert ls == int(g["level"][b][i])
            np.testing.assert_allclose(ss, so, rtol=0, atol=1e-8 * so[0])  # north star: 1e-4 rel
            np.testing.assert_allclose(Bs[:, idx].T @ Bs[:, idx], g["gram_block"][b][i], rtol=0, atol=1e-8 * so[0] ** 2)
            assert abs(ds - float(g["delta"][b][i])) <= 1e-8 * so[0] ** 2
    lanes.close()
    for sk in singles:
        sk.close()


def test_rsvd_rank_deficient_adjacency_matches_oracle(eng):
    """Adjacency of rank < n_components + 10 (only 20 of 400 rows have neighbours): the Cholesky-QR normaliser meets zero
    pivots, its weak-pivot flag sends the final basis through the Householder chain; singular values as the oracle's
    (= the reference's LU / QR chain), zeros included."""
    from mused_amd import matrix_operations as mo
    from oracle import mo_oracle as omo

    rng = np.random.default_rng(5)
    Xs = rng.standard_normal((400, 6))
    Xs[rng.permutation(400)[20:], 0] = np.nan  # rows without a valid neighbourhood (matrix_operations.py:114-115)
    A = omo.create_adjacency_matrix(Xs, "", 5)
    assert np.linalg.matrix_rank(A) <= 20 < 30
    _, sig_o, _ = omo.randomized_svd_reduce(A, 20, 3)
    adj = mo.adjacency_on_device(Xs, "", 5, engine=eng)
    _, sig_d = eng.svd_reduce(adj, 20, 3)
    np.testing.assert_allclose(sig_d.cpu().numpy(), sig_o, rtol=0, atol=1e-8 * sig_o[0])
    assert eng.rsvd_fallbacks >= 1  # the window went through the LU / Householder handle
    # and a full-rank window right after it on the same handle (flag cleared per call)
    fb = eng.rsvd_fallbacks
    X = rng.standard_normal((400, 12))
    A2 = omo.create_adjacency_matrix(X, "", 10)
    _, s2o, _ = omo.randomized_svd_reduce(A2, 20, 3)
    _, s2d = eng.svd_reduce(mo.adjacency_on_device(X, "", 10, engine=eng), 20, 3)
    np.testing.assert_allclose(s2d.cpu().numpy(), s2o, rtol=1e-9)
    assert eng.rsvd_fallbacks == fb
    # the in-graph fallback (mode 0, what a bare C-ABI caller gets) on the rank-deficient window
    from mused_amd.engine import WindowEngine

    e0 = WindowEngine(400)
    e0.rsvd_mode = "graph"
    _, sig_g = e0.svd_reduce(mo.adjacency_on_device(Xs, "", 5, engine=e0), 20, 3)
    np.testing.assert_allclose(sig_g.cpu().numpy(), sig_o, rtol=0, atol=1e-8 * sig_o[0])
    e0.close()


def test_rsvd_edge_overflow_is_flagged_and_memory_safe(eng):
    """More edges than nnz_cap (cosine selects k + 1 per row; a zero row is not its own nearest): the neighbour lists
    are truncated inside their buffer, flags[0] is raised and the host check raises; with the right cap it is clean."""
    from mused_amd._lib import MusedError
    from mused_amd.engine import WindowEngine

    rng = np.random.default_rng(3)
    X = rng.standard_normal((600, 16)).astype(np.float32)
    X[::7] = 0.0  # zero-norm rows: cosine 0 to every row, ties to the lowest columns -> k + 1 edges
    k = 9
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), k, "cosine")
    nnz = int(adj.degrees()[2][1].item())
    assert nnz > 600 * k
    e2 = WindowEngine(600)
    emb, sig, flags = e2.svd_reduce(adj, 4, 0, nnz_cap=600 * k, want_flags=True)  # too small on purpose
    torch.cuda.synchronize()
    assert int(flags.cpu()[0]) != 0
    with pytest.raises(MusedError):
        WindowEngine.check_rsvd_flags(flags.cpu().numpy())
    e2.close()
    e3 = WindowEngine(600)
    emb, sig, flags = e3.svd_reduce(adj, 4, 0, nnz_cap=600 * (k + 1), want_flags=True)
    assert int(flags.cpu()[0]) == 0 and bool(torch.isfinite(emb).all())
    e3.close()


def test_full_size_swfd_properties():
    """d = 1024, l = 128, N = 10,000 (config 2): covariance error bound against the exact window Gram
    (computed on the device), orthogonal sketch rows, and the sketch forgets an expired window."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD as Dev

    W, d, ell = 10000, 1024, 128
    Xs = [torch.from_numpy(synth.stream_window("blob", t, W, d, 0)[0]).cuda() for t in range(2)]
    R = float((Xs[0].double() ** 2).sum(1).max().item())
    sk = Dev(N=W, R=R, d=d, sketch_dim=ell)
    for t in range(2):
        sk.fit(Xs[t])
        B, sig, info = sk.get_device()
        A = Xs[t].double()
        E = A.t() @ A - B.t() @ B
        err = torch.linalg.matrix_norm(E, ord=2).item()
        f2 = (A * A).sum().item()
        assert err <= 1.0 * f2 / ell, (t, err, f2 / ell)  # same bound as the oracle property test
        Gs = B @ B.t()
        off = (Gs - torch.diag(torch.diag(Gs))).abs().max().item()
        assert off <= 1e-8 * Gs.max().item()
        assert torch.allclose(sig, B.norm(dim=1), rtol=1e-12)
    sk.close()


def test_config3_shapes_l256():
    """BASELINE config 3 uses l = 256: the 512- and 1024-order eigenproblems of the sketch and the 266-column
    panels of the eigenstep, at a window small enough for the CPU oracle."""
    from mused_amd import synth
    from mused_amd.engine import WindowEngine
    from oracle import mo_oracle as omo

    N, ell, d = 700, 256, 320
    X, _ = synth.make_stream("blob", N + 300, d, 11)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    dev, ora = _swfd_pair(N, R, d, ell)
    for lo, hi in [(0, 300), (300, 700), (700, 1000)]:
        dev.fit(X[lo:hi])
        ora.fit(X[lo:hi])
        _compare(dev, ora, f"l=256 t={hi}")
    dev.close()
    # eigenstep with reduced_dim = 256 on a 1500-row window (r = 266 random columns)
    Xw, _ = synth.blob_stream(1500, 96, 3, n_centres=6)
    eng = WindowEngine(1500)
    adj = eng.knn_adjacency(torch.from_numpy(Xw).cuda(), 40)
    emb, sig = eng.svd_reduce(adj, 256, 0, nnz_cap=1500 * 40)
    A = omo.create_adjacency_matrix(Xw, "", 40)
    e_ref, s_ref, _ = omo.randomized_svd_reduce(A, 256, 0)
    np.testing.assert_allclose(sig.cpu().numpy(), s_ref, rtol=1e-8)
    np.testing.assert_allclose(emb.cpu().numpy(), e_ref, atol=1e-6 * np.abs(e_ref).max())
    eng.close()


@pytest.mark.parametrize("d", [160, 96])
def test_rank_deficient_buffers(d):
    """Sketch buffers whose Gram matrices are dense AND rank deficient (row length d < 2l, and d < l): the
    null-space columns of the one-sided Jacobi need about twice the sweeps of a full-rank buffer before they
    fall below the drop tolerance -- the adaptive sweep count has to notice (a fixed count tuned on full-rank
    buffers was 1e-5 .. 1e-2 off here)."""
    from mused_amd import synth

    N, ell = 700, 128
    X, _ = synth.make_stream("blob", N + 300, d, 11)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    dev, ora = _swfd_pair(N, R, d, ell)
    for lo, hi in [(0, 300), (300, 700), (700, 1000)]:
        dev.fit(X[lo:hi])
        ora.fit(X[lo:hi])
        _compare(dev, ora, f"d={d} t={hi}")
    dev.close()


def test_duplicate_levels_solved_once_changes_nothing(monkeypatch):
    """Sketch levels that are still bit-identical are solved once (mused_swfd: representative map); switching that
    off must give the same sketch, snapshot rings and exported state bit for bit."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD as Dev

    N, ell, d = 600, 16, 64
    X, _ = synth.make_stream("blob", 3 * N + 77, d, 5)
    R = float((X.astype(np.float64) ** 2).sum(1).max())
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MUSED_SWFD_DEDUPE", flag)
        sk = Dev(N=N, R=R, d=d, sketch_dim=ell)
        res = []
        for lo, hi in [(0, 250), (250, 1300), (1300, 3 * N + 77)]:
            sk.fit(X[lo:hi])
            B, s, lvl, dl = sk.get()
            res.append((B.copy(), s.copy(), lvl, dl))
        res.append(sk.export_half(1).cpu().numpy().copy())
        res.append(sk.export_half(0).cpu().numpy().copy())
        sk.close()
        outs.append(res)
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][4], outs[1][4])
