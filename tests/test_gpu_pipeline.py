"""End-to-end on the GPU: the window loop (main.py:13-130 role) and the NumPy drop-in call surface."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import load_golden, regen_inputs

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize(
    "name", ["c1_stream_blob_s0", "c1_stream_blob_s1", "c1_stream_gauss_s0", "c4s_stream_twomod_s0"]
)
@pytest.mark.parametrize("async_labels", [True, False])
def test_stream_event_labels_bit_exact(name, async_labels):
    """Whole-run `all_clusters` of the reference (approach sSVDMC) reproduced bit for bit by the device
    pipeline, one and two modalities, with and without the asynchronous label worker."""
    from mused_amd.pipeline import StreamPipeline

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    pipe = StreamPipeline(W, ell, k, seed, "sSVDMC", async_labels=async_labels)
    out = pipe.run(mods, labels)
    pipe.close()
    assert np.array_equal(out.astype(np.int64), g["all_clusters"])
    assert hashlib.sha256(out.astype(np.int64).tobytes()).hexdigest() == str(g["labels_sha"])


def _bench_stream(g, kind="blob"):
    from mused_amd import synth

    n_windows, W, d, ell, k, seed = (int(x) for x in g["meta"])
    wins = [synth.stream_window(kind, t, W, d, seed) for t in range(n_windows)]
    assert [synth.array_digest(w[0]) for w in wins] == [str(x) for x in g["window_digest"]]
    return np.concatenate([w[0] for w in wins]), np.concatenate([w[1] for w in wins]), (n_windows, W, d, ell, k, seed)


@pytest.mark.parametrize("approach", ["sSVDMC", "SWFDMC"])
def test_reference_default_parameters_stream(approach):
    """The reference's own operating point (/root/reference/main.py:305-313: window_size 2000, reduced_dim 50, k_basis 50).
    sSVDMC: ten windows, `all_clusters` of the reference's window loop (golden bench_refdef_blob_s0) bit for bit -- the
    eigenstep's r = 60 Gram on the direct solver.  SWFDMC: three windows of the same stream through the reference's wiring of
    the sketch (rotations of order 100, queries of order 150, all on the direct solver) against the ORACLE pipeline's fixture
    swfdmc_refdef_3win (device == specification: the reference's swfd submodule is absent, parity unpinned)."""
    from mused_amd import synth
    from mused_amd.pipeline import StreamPipeline

    if approach == "sSVDMC":
        g = load_golden("bench_refdef_blob_s0")
        X, labels, (n_windows, W, d, ell, k, seed) = _bench_stream(g)
        with StreamPipeline(W, ell, k, seed, approach, modality_types=[""], async_labels=False) as pipe:
            out = np.asarray(pipe.run([X.astype(np.float64)], labels), dtype=np.int64)
        assert np.array_equal(out.reshape(n_windows, W), g["labels"].astype(np.int64))
        assert hashlib.sha256(out.tobytes()).hexdigest()[:16] == str(g["cumulative_sha16"][n_windows - 1])
        return
    g = load_golden("swfdmc_refdef_3win")
    W, ell, k, seed, n_windows, d = (int(x) for x in g["meta"][:6])
    assert (W, ell, k) == (2000, 50, 50)
    wins = [synth.stream_window("blob", t, W, d, seed) for t in range(n_windows)]
    X = np.concatenate([w[0] for w in wins])
    labels = np.concatenate([w[1] for w in wins])
    assert [synth.array_digest(X)] == [str(x) for x in g["input_digest"]]
    with StreamPipeline(W, ell, k, seed, approach, modality_types=[""], async_labels=False) as pipe:
        out = np.asarray(pipe.run([X.astype(np.float64)], labels), dtype=np.int64)
        for tr, sig in zip(pipe.trace, g["sigma"]):
            np.testing.assert_allclose(tr["sigma"], sig, rtol=0, atol=1e-8 * sig[0])
    assert np.array_equal(out, g["all_clusters"].astype(np.int64))


@pytest.mark.parametrize("name,ratio", [("c1_stream_hop2_blob_s0", 2), ("c1_stream_hop4_gauss_s1", 4)])
def test_hopping_windows_match_reference_golden(name, ratio):
    """step_window_ratio = 2 and 4 (main.py:32) against fixtures produced by the reference's own window loop."""
    from mused_amd.pipeline import StreamPipeline

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    pipe = StreamPipeline(W, ell, k, seed, "sSVDMC", step_window_ratio=ratio)
    out = pipe.run(mods, labels)
    pipe.close()
    assert np.array_equal(out.astype(np.int64), g["all_clusters"])


def test_process_streaming_data_signature_and_hopping_windows():
    """Same positional parameters as main.py:13; step_window_ratio = 2 (hopping windows, main.py:32)
    against the oracle's window loop."""
    from mused_amd import synth
    from mused_amd.pipeline import process_streaming_data
    from oracle import mo_oracle as omo

    X, labels = synth.blob_stream(1200, 24, 2, n_centres=3)
    res = process_streaming_data({}, [X.astype(np.float64)], [""], 400, 8, 20, 3, 0, "sSVDMC", labels, 2, 0.0, "types", False, 1.5, 2)
    ref = omo.process_streaming_data([X.astype(np.float64)], [""], 400, 8, 20, 0, "sSVDMC", labels, step_window_ratio=2)
    assert len(ref) == 5 * 400  # triggers at rows 400, 600, ..., 1200
    assert np.array_equal(res["all_clusters"], ref)
    assert res["processing_time"] > 0


def test_stream_with_nonfinite_rows_and_modality_types():
    """The drop-in window loop filters non-finite rows like the reference (matrix_operations.py:114-115), runs the
    "text" modality (host TF-IDF + device cosine) next to a numeric one, and the five SED2012 modality types together
    (data_loader.py:113) with labels equal to the CPU oracle's window loop."""
    from mused_amd import synth
    from mused_amd.pipeline import process_streaming_data
    from oracle import mo_oracle as omo

    X, labels = synth.blob_stream(900, 20, 4, n_centres=3)
    X = X.astype(np.float64)
    X[5, 3] = np.nan
    X[301, 0] = np.inf
    X[777, 19] = -np.inf
    res = process_streaming_data({}, [X], [""], 300, 6, 15, 3, 0, "sSVDMC", labels, 1, 0.0, "types", False, 1.5, 2)
    ref = omo.process_streaming_data([X], [""], 300, 6, 15, 0, "sSVDMC", labels)
    assert np.array_equal(res["all_clusters"], ref)
    text, tl = synth.text_stream(600, 1)
    Xn = synth.blob_stream(600, 12, 1, n_centres=4)[0].astype(np.float64)
    res = process_streaming_data({}, [Xn, text], ["", "text"], 300, 6, 15, 4, 0, "sSVDMC", tl, 1, 0.0, "types", False, 1.5, 2)
    ref = omo.process_streaming_data([Xn, text], ["", "text"], 300, 6, 15, 0, "sSVDMC", tl)
    assert np.array_equal(res["all_clusters"], ref)
    cols, ml = synth.metadata_stream(900, 7)
    text, _ = synth.text_stream(900, 7)
    mods = [cols["location"], cols["time"], cols["username"], cols["tags"], text]
    types_ = ["location", "time", "username", "tags", "text"]
    res = process_streaming_data({}, mods, types_, 300, 8, 8, 5, 0, "sSVDMC", ml, 1, 0.0, "types", False, 1.5, 2)
    ref = omo.process_streaming_data(mods, types_, 300, 8, 8, 0, "sSVDMC", ml)
    assert np.array_equal(res["all_clusters"], ref)


def test_window_slots_reproduce_the_reference_labels():
    """Consecutive windows on several engines / streams (window_slots) -- the label chain still sees them in order."""
    from conftest import load_golden, regen_inputs
    from mused_amd.pipeline import StreamPipeline

    g = load_golden("c1_stream_blob_s0")
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    for slots in (2, 3):
        pipe = StreamPipeline(W, ell, k, seed, "sSVDMC", modality_types=[""], window_slots=slots)
        out = pipe.run([m.astype(np.float64) for m in mods], labels)
        pipe.close()
        assert np.array_equal(np.asarray(out, dtype=np.int64), g["all_clusters"])
    g = load_golden("metadata")  # string / metadata modality types through the slots as well
    n, W, ell, k, seed, sseed = (int(x) for x in g["run_meta"])
    from mused_amd import synth

    cols, labels = synth.metadata_stream(n, sseed)
    pipe = StreamPipeline(W, ell, k, seed, "sSVDMC", modality_types=["location", "username"], window_slots=3)
    out = pipe.run([cols["location"], cols["username"]], labels)
    pipe.close()
    assert np.array_equal(np.asarray(out, dtype=np.int64), g["run_clusters"])


def test_metadata_run_matches_reference_golden():
    """Whole run of the reference's window loop over the (location, username) columns (tests/golden/metadata.npz)."""
    from conftest import load_golden
    from mused_amd import synth
    from mused_amd.pipeline import process_streaming_data

    g = load_golden("metadata")
    n, W, ell, k, seed, sseed = (int(x) for x in g["run_meta"])
    cols, labels = synth.metadata_stream(n, sseed)
    res = process_streaming_data({}, [cols["location"], cols["username"]], ["location", "username"], W, ell, k,
                                 len(np.unique(labels)), seed, "sSVDMC", labels, 1, 0.0, "types", False, 1.5, 2)
    assert np.array_equal(np.asarray(res["all_clusters"], dtype=np.int64), g["run_clusters"])


def test_swfdmc_approach_matches_oracle_pipeline():
    """approach SWFDMC (main.py:58-76): SWFD over the rows of the fused matrix (d = W), R from the first
    window only, sketch transposed to (W, l), labels equal to the CPU oracle pipeline."""
    from mused_amd import synth
    from mused_amd.pipeline import StreamPipeline
    from oracle import mo_oracle as omo
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    mods, labels = synth.two_modality_blob_stream(3 * 256, 16, 5, n_centres=3)
    trace = []
    ref = omo.process_streaming_data([m.astype(np.float64) for m in mods], ["", ""], 256, 6, 12, 0, "SWFDMC", labels,
                                     swfd_cls=OraSWFD, trace=trace)
    pipe = StreamPipeline(256, 6, 12, 0, "SWFDMC", async_labels=False)
    out = pipe.run(mods, labels)
    for a, b in zip(pipe.trace, trace):
        np.testing.assert_allclose(a["sigma"], b["sigma"], rtol=0, atol=1e-8 * b["sigma"][0])
    pipe.close()
    assert np.array_equal(out, ref)


def test_swfdmc_bitmask_rows_at_w2000_two_modalities():
    """SWFDMC in the reference's wiring at a window of 2,000 rows, two modalities: the sketch is fed the rows of the
    fused adjacency straight from the device bitmask (no dense W x W matrix), bit-identical to feeding the dense int64
    matrix, and the event labels equal the CPU oracle pipeline's."""
    from mused_amd import synth
    from mused_amd.engine import WindowEngine
    from mused_amd.pipeline import StreamPipeline
    from mused_amd.swfd import SeqBasedSWFD
    from oracle import mo_oracle as omo
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    W, ell, k = 2000, 16, 20
    mods, labels = synth.two_modality_blob_stream(2 * W, 24, 9, n_centres=4)
    # (1) bit rows == dense rows, bit for bit
    eng = WindowEngine(W)
    fused = eng.fuse([eng.knn_adjacency(torch.from_numpy(m[:W]).cuda(), k) for m in mods])
    R = eng.max_row_sq_norm(fused)
    a, b = SeqBasedSWFD(N=W, R=R, d=W, sketch_dim=ell), SeqBasedSWFD(N=W, R=R, d=W, sketch_dim=ell)
    a.fit_adjacency(fused)
    b.fit(fused.to_dense())
    Ba, sa, la, da = a.get()
    Bb, sb, lb, db = b.get()
    assert la == lb and da == db and np.array_equal(Ba, Bb) and np.array_equal(sa, sb)
    a.close(), b.close(), eng.close()
    # (2) whole pipeline against the oracle pipeline
    trace = []
    ref = omo.process_streaming_data([m.astype(np.float64) for m in mods], ["", ""], W, ell, k, 0, "SWFDMC", labels,
                                     swfd_cls=OraSWFD, trace=trace)
    pipe = StreamPipeline(W, ell, k, 0, "SWFDMC", async_labels=False)
    out = pipe.run(mods, labels)
    for x, y in zip(pipe.trace, trace):
        np.testing.assert_allclose(x["sigma"], y["sigma"], rtol=0, atol=1e-8 * y["sigma"][0])
    pipe.close()
    assert np.array_equal(out, ref)


def test_numpy_dropin_call_surface():
    """`import matrix_operations` / `from swfd import SeqBasedSWFD` via mused_amd/compat, NumPy in/out."""
    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "mused_amd", "compat"))
    try:
        import matrix_operations as mo  # the drop-in, NOT the reference (which is not on sys.path here)
        from swfd import SeqBasedSWFD

        assert "mused_amd" in mo.create_adjacency_matrix.__module__
        g = load_golden("edges")
        A = mo.create_adjacency_matrix(g["nonfinite_X"], "", 5)  # non-finite rows dropped
        assert A.dtype == np.float64 and np.array_equal(A.astype(np.uint8), g["nonfinite_A"])
        X = g["k1_X"]
        assert np.array_equal(mo.create_adjacency_matrix(X, "", 1).astype(np.uint8), g["k1_A"])
        assert np.array_equal(mo.create_adjacency_matrix(X, "", 0).astype(np.uint8), g["k0_A"])
        assert np.array_equal(mo.create_adjacency_matrix(X, "", 12).astype(np.uint8), g["kn_A"])
        with pytest.raises(ValueError):
            mo.create_adjacency_matrix(X, "", 13)
        A1, A2 = g["fuse_A1"].astype(np.float64), g["fuse_A2"].astype(np.float64)
        F2 = mo.fuse_matrices([A1, A2])
        assert str(F2.dtype) == str(g["fuse2_dtype"]) and np.array_equal(F2.astype(np.uint8), g["fuse2"])
        F1 = mo.fuse_matrices([A1])
        assert str(F1.dtype) == str(g["fuse1_dtype"]) and np.array_equal(F1, A1)
        with pytest.raises(NotImplementedError):
            mo.perform_svd_reduction(np.full((4, 4), 0.5), 2, 0)
        sk = SeqBasedSWFD(N=8, R=1.0, d=8, sketch_dim=2)  # main.py:62 keywords, demo sizes
        out = sk.get()
        assert len(out) == 4 and out[0].shape == (2, 8) and not out[0].any()
        rng = np.random.default_rng(0)
        for _ in range(20):
            sk.fit(rng.integers(0, 2, size=(1, 8)))
        assert sk.get()[0].shape == (2, 8)
        with pytest.raises(ValueError):
            sk.fit(np.zeros((1, 7)))
    finally:
        sys.path.pop(0)
        sys.modules.pop("matrix_operations", None)
        sys.modules.pop("swfd", None)
