"""hopping windows (step_window_ratio > 1, main.py:32) reuse the similarity work of the previous window
(mused_knn_fused_hop) -- bit-identical to computing every window from scratch."""
import numpy as np
import pytest

from conftest import load_golden, regen_inputs

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("W,hop,d,k,metric,dtype", [
    (1024, 512, 24, 20, "l2", np.float64),     # ratio 2, tile aligned
    (1000, 250, 16, 50, "l2", np.float32),     # ratio 4, nothing aligned to the 128-row tiles
    (1500, 300, 32, 30, "cosine", np.float64),  # ratio 5, cosine
    (700, 699, 8, 10, "l2", np.float64),       # almost everything new
    (900, 1, 12, 25, "l2", np.float64),        # one row per hop
])
def test_reuse_equals_recompute_bitwise(W, hop, d, k, metric, dtype):
    from mused_amd import synth
    from mused_amd.engine import WindowEngine

    nwin = 7
    X, _ = synth.blob_stream(W + hop * nwin, d, 3, n_centres=5)
    X = X.astype(dtype)
    Xd = torch.from_numpy(X).cuda()
    eng, fresh = WindowEngine(W), WindowEngine(W)
    for t in range(nwin + 1):
        lo = t * hop
        rows = Xd[lo:lo + W]
        a = eng.knn_adjacency_hop(rows, k, metric, key=0, lo=lo)
        b = fresh.knn_adjacency(rows, k, metric)
        assert torch.equal(a.mask, b.mask), f"window {t} (rows {lo}..{lo + W}) differs"
    assert eng.hop_windows == nwin + 1 and eng.hop_reused == nwin and eng.hop_recomputes == 0
    eng.close(), fresh.close()


def test_reuse_survives_a_distribution_shift_and_gaps():
    """Rows far from everything seen so far enter the window (the kept thresholds of the staying rows prove nothing about
    them: they only matter for the entering rows' own lists), the stream then jumps (no overlap: computed from scratch),
    and a window is skipped (overlap with the window before the last one)."""
    from mused_amd.engine import WindowEngine

    rng = np.random.default_rng(0)
    W, d, k = 800, 10, 15
    X = rng.standard_normal((6000, d))
    X[2000:2400] *= 0.01          # a very tight cluster arrives: every staying row's neighbours change
    X[3000:3300] += 50.0          # and a far one
    Xd = torch.from_numpy(X).cuda()
    eng, fresh = WindowEngine(W), WindowEngine(W)
    for lo in [0, 200, 400, 1200, 1400, 1800, 2000, 2200, 2400, 2600, 2800, 3000, 3200, 4800, 5000, 5200]:
        rows = Xd[lo:lo + W]
        a = eng.knn_adjacency_hop(rows, k, "l2", key="m", lo=lo)
        b = fresh.knn_adjacency(rows, k, "l2")
        assert torch.equal(a.mask, b.mask), f"window at {lo} differs"
    assert eng.hop_reused >= 8
    eng.close(), fresh.close()


@pytest.mark.parametrize("name,ratio", [("c1_stream_hop2_blob_s0", 2), ("c1_stream_hop4_gauss_s1", 4)])
def test_pipeline_reuses_and_matches_reference_golden(monkeypatch, name, ratio):
    """The window loop with step_window_ratio 2 / 4 against the reference's own labels, with the reuse on (the default
    from ratio 4; forced here for ratio 2 as well) -- and the engine reports that it reused."""
    from mused_amd.pipeline import StreamPipeline

    monkeypatch.setenv("MUSED_HOP_REUSE", "1")

    g = load_golden(name)
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    with StreamPipeline(W, ell, k, seed, "sSVDMC", step_window_ratio=ratio) as pipe:
        out = pipe.run(mods, labels)
        reused, windows = pipe.eng.hop_reused, pipe.eng.hop_windows
    assert np.array_equal(out.astype(np.int64), g["all_clusters"])
    assert windows >= 2 and reused == windows - 1


def test_flagged_hop_state_is_reset_by_its_key_not_by_its_slot(monkeypatch):
    """the deferred flag words are indexed by the window's DEFERRED kNN calls, the hop states by modality.  With
    modality types ["text", ""] the text modality takes no flag slot, so the numeric modality's hop state (key 1) reports
    in slot 0.  A raised word must reset THAT state (round 3 looked for key 0, found nothing, and the sticky flag repeated
    every later window on the fallback engine)."""
    from mused_amd import synth
    from mused_amd.engine import WindowEngine
    from mused_amd.pipeline import StreamPipeline
    from oracle import mo_oracle as omo

    monkeypatch.setenv("MUSED_HOP_REUSE", "1")
    W, ratio, n = 300, 3, 1500
    text, labels = synth.text_stream(n, 5)
    X, _ = synth.blob_stream(n, 12, 5, n_centres=4)
    X = X.astype(np.float64)
    real = WindowEngine.window_flags
    calls = {"n": 0, "reset": []}

    def flags_with_one_raised(self, rsvd_flags):
        out = real(self, rsvd_flags)
        calls["n"] += 1
        if calls["n"] == 2:          # second window: pretend its first deferred kNN call failed its certificate
            out[4] = 1
        return out

    real_req = WindowEngine.request_hop_reset

    def spy(self, key):
        calls["reset"].append(key)
        return real_req(self, key)

    monkeypatch.setattr(WindowEngine, "window_flags", flags_with_one_raised)
    monkeypatch.setattr(WindowEngine, "request_hop_reset", spy)
    with StreamPipeline(W, 8, 10, 0, "sSVDMC", modality_types=["text", ""], step_window_ratio=ratio) as pipe:
        out = pipe.run([text, X], labels)
        redone, reused, windows = pipe.redone_windows, pipe.eng.hop_reused, pipe.eng.hop_windows
    ref = omo.process_streaming_data([text, X], ["text", ""], W, 8, 10, 0, "sSVDMC", labels, step_window_ratio=ratio)
    assert np.array_equal(np.asarray(out, dtype=np.int64), np.asarray(ref, dtype=np.int64))
    assert calls["reset"] == [1]                      # the numeric modality's state, found through the slot -> key map
    assert redone == 1                                # only the flagged window was repeated
    assert windows >= 4 and reused <= windows - 2     # ... and one later window was rebuilt from scratch
