"""The sketch at the full sizes the smaller parity tests do not reach: BASELINE config 3's shape (d = 4096, l = 256:
order-512 rotations, order-768 query) against the oracle, and the SWFDMC approach in the reference's wiring
(main.py:58-76: the sketch over the rows of the fused W x W adjacency) at the headline window W = 10,000."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_swfd_config3_orders_past_two_epoch_ends():
    """d = 4096, l = 256 (BASELINE config 3's orders: rotations of order 512 on the blocked direct solver, queries of order
    768 / 1024) with a reduced window N = 1,024 over 2,500 rows: dumps, two epoch ends with the AUX -> MAIN swap, expiry --
    device == specification after every ragged block, against tests/golden/swfd_c3orders.npz (outputs of oracle/swfd_oracle.py,
    make_swfd_fixtures.py; the reference's swfd submodule is absent: parity unpinned)."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD

    g = load_golden("swfd_c3orders")
    N, d, ell, seed = (int(x) for x in g["meta"][:4])
    steps = [int(x) for x in g["meta"][4:]]
    X, _ = synth.stream_window("blob", 0, sum(steps), d, seed)
    assert synth.array_digest(X) == str(g["input_digest"])
    Xd = torch.from_numpy(X.astype(np.float64)).cuda()
    dev = SeqBasedSWFD(N=N, R=float(g["R"]), d=d, sketch_dim=ell)
    idx = g["gram_idx"]
    t = 0
    for i, step in enumerate(steps):          # crosses row 1024 and row 2048 (epoch ends), ragged in between
        dev.fit(Xd[t:t + step])
        t += step
        Bd, sd, ld, dd = dev.get()
        so = g["sigma"][i]
        assert ld == int(g["level"][i]), t
        np.testing.assert_allclose(sd, so, rtol=0, atol=1e-8 * so[0], err_msg=str(t))
        np.testing.assert_allclose(dd, float(g["delta"][i]), rtol=1e-9, atol=1e-9 * so[0] ** 2)
        np.testing.assert_allclose(Bd[:, idx].T @ Bd[:, idx], g["gram_block"][i], rtol=0, atol=1e-8 * so[0] ** 2, err_msg=str(t))
    dev.close()


def test_swfd_at_config3_shape_matches_oracle():
    """N = 10,000, d = 4096, l = 256 (BASELINE config 3): 640 rows = two full rotations + a ragged tail, then get():
    singular values, level, final shrink and a sampled block of the covariance against the CPU specification
    (tests/golden/swfd_c3shape.npz, make_swfd_fixtures.py), 1e-8 sigma_1 (the north star asks 1e-4 relative).
    Device == specification, not reference (parity unpinned)."""
    from mused_amd import synth
    from mused_amd.swfd import SeqBasedSWFD

    g = load_golden("swfd_c3shape")
    N, d, ell, seed = (int(x) for x in g["meta"][:4])
    rows = int(g["meta"][4])
    X, _ = synth.stream_window("blob", 0, rows, d, seed)
    assert synth.array_digest(X) == str(g["input_digest"])
    dev = SeqBasedSWFD(N=N, R=float(g["R"]), d=d, sketch_dim=ell)
    dev.fit(torch.from_numpy(X.astype(np.float64)).cuda())
    Bd, sd, ld, dd = dev.get()
    dev.close()
    so, idx = g["sigma"][0], g["gram_idx"]
    assert ld == int(g["level"][0])
    np.testing.assert_allclose(sd, so, rtol=0, atol=1e-8 * so[0])
    np.testing.assert_allclose(dd, float(g["delta"][0]), rtol=1e-9, atol=1e-9 * so[0] ** 2)
    np.testing.assert_allclose(Bd[:, idx].T @ Bd[:, idx], g["gram_block"][0], rtol=0, atol=1e-8 * so[0] ** 2)


@pytest.mark.parametrize("tag", ["swfdmc_w10k_m1", "swfdmc_w10k_m2", "swfdmc_w10k_m1_3win"])
def test_swfdmc_reference_wiring_at_w10000(tag):
    """approach SWFDMC at W = 10,000 (one and two modalities): R from the first window, the sketch fed the 10,000 bit
    rows of the fused adjacency (d = W), get() transposed to (W, l), k-means, matching -- singular values and event
    labels against the oracle pipeline's fixture (tests/golden/make_swfd_fixtures.py).  `_3win`: three windows -- the
    AUX -> MAIN swap at both epoch starts, the expiry of the first window's snapshots and two Hungarian steps at d = W =
    10,000.  These fixtures pin device == THIS REPO'S SPECIFICATION of the sketch (oracle/swfd_oracle.py), not the
    reference: its swfd submodule is absent (parity unpinned)."""
    from mused_amd import synth
    from mused_amd.pipeline import StreamPipeline

    g = load_golden(tag)
    W, ell, k, seed, n_windows = (int(x) for x in g["meta"][:5])
    dims = tuple(int(x) for x in g["meta"][5:])
    wins = [synth.stream_window("blob", t, W, dims[0], seed) if len(dims) == 1 else synth.stream_window_mods(t, W, dims, seed)
            for t in range(n_windows)]
    X = np.concatenate([w[0] for w in wins])
    labels = np.concatenate([w[1] for w in wins])
    mods, c0 = [], 0
    for dm in dims:
        mods.append(np.ascontiguousarray(X[:, c0 : c0 + dm]))
        c0 += dm
    assert [synth.array_digest(m) for m in mods] == [str(x) for x in g["input_digest"]]
    with StreamPipeline(W, ell, k, seed, "SWFDMC", modality_types=[""] * len(mods), async_labels=False) as pipe:
        out = pipe.run([m.astype(np.float64) for m in mods], labels)
        for tr, sig in zip(pipe.trace, g["sigma"]):
            np.testing.assert_allclose(tr["sigma"], sig, rtol=0, atol=1e-8 * sig[0])
    assert np.array_equal(np.asarray(out, dtype=np.int64), g["all_clusters"].astype(np.int64))
    assert hashlib.sha256(np.asarray(out, dtype=np.int64).tobytes()).hexdigest() == str(g["labels_sha"])


@pytest.mark.parametrize("n_lanes", [2, 3])
def test_swfdmc_lanes_equal_the_sequential_specification(n_lanes):
    """SWFDMC with the three fixture windows dealt to 2 / 3 lock-step lanes (each block preceded by its halo window, a window of
    empty rows at the beginning of the stream; mused_amd.pipeline.SwfdmcLanes): singular values and event labels equal the
    SEQUENTIAL oracle pipeline's fixture -- device lanes == specification (not reference: parity unpinned)."""
    from mused_amd import synth
    from mused_amd.pipeline import SwfdmcLanes

    g = load_golden("swfdmc_w10k_m1_3win")
    W, ell, k, seed, n_windows = (int(x) for x in g["meta"][:5])
    d = int(g["meta"][5])
    wins = [synth.stream_window("blob", t, W, d, seed) for t in range(n_windows)]
    assert [synth.array_digest(np.concatenate([w[0] for w in wins]))] == [str(x) for x in g["input_digest"]]
    windows = [[torch.from_numpy(w[0].astype(np.float64)).cuda()] for w in wins]
    labels = [w[1] for w in wins]
    R = SwfdmcLanes.r_of_first_window(windows[0], W, k)
    with SwfdmcLanes(W, ell, k, seed, n_lanes, R, modality_types=[""]) as lanes:
        out = lanes.run(windows, labels)
        for t, sig in enumerate(g["sigma"]):
            np.testing.assert_allclose(lanes.sigma[t], sig, rtol=0, atol=1e-8 * sig[0])
    assert np.array_equal(np.asarray(out, dtype=np.int64), g["all_clusters"].astype(np.int64))
    assert hashlib.sha256(np.asarray(out, dtype=np.int64).tobytes()).hexdigest() == str(g["labels_sha"])
