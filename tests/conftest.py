import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: several seconds of CPU work")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not generated")
    return np.load(path, allow_pickle=False)


def nbr_hash(A) -> str:
    """Same digest as tests/golden/make_golden.py: rows of sorted neighbour columns."""
    A = np.asarray(A)
    h = hashlib.sha256()
    for i in range(A.shape[0]):
        cols = np.flatnonzero(A[i]).astype(np.int32)
        h.update(np.int32(i).tobytes())
        h.update(np.int32(len(cols)).tobytes())
        h.update(cols.tobytes())
    return h.hexdigest()


def nbr_hash_from_lists(rows) -> str:
    """Digest from per-row sorted int32 neighbour arrays (no dense matrix)."""
    h = hashlib.sha256()
    for i, cols in enumerate(rows):
        cols = np.asarray(cols, dtype=np.int32)
        h.update(np.int32(i).tobytes())
        h.update(np.int32(len(cols)).tobytes())
        h.update(cols.tobytes())
    return h.hexdigest()


def regen_inputs(g):
    """Regenerate the seeded inputs of a golden case and check their digest."""
    from mused_amd import synth

    n, d, W, ell, k, seed = (int(x) for x in g["meta"])
    kind = str(g["kind"])
    if kind == "blob2":
        mods, labels = synth.two_modality_blob_stream(n, d, seed, n_centres=int(g["n_centres"]) if "n_centres" in g.files else 4)
    elif kind == "blob":
        nc = 8 if d >= 256 else 4
        X, labels = synth.blob_stream(n, d, seed, n_centres=nc, sep=2.0)
        mods = [X]
    else:
        X, labels = synth.make_stream(kind, n, d, seed)
        mods = [X]
    digests = [synth.array_digest(m) for m in mods]
    assert digests == [str(x) for x in g["input_digest"]], "regenerated inputs differ from the golden run's"
    return mods, labels, (n, d, W, ell, k, seed)


def text_inputs(g):
    """Regenerate the synthetic string records of tests/golden/cosine.npz and check their digest."""
    from mused_amd import synth

    n, k, seed = (int(x) for x in g["text_meta"])
    data, labels = synth.text_stream(n, seed)
    assert hashlib.sha256("\x1f".join(data.ravel()).encode()).hexdigest() == str(g["text_digest"])
    return data, labels, n, k


METADATA_TYPES = ("location", "time", "username", "tags")
# reference picks that are decided by its unstable argsort between EQUAL scores (matrix_operations.py:53, 88): every
# "tags" adjacency (most Jaccard similarities are 0) and the "time" adjacency of the whole-hour stream B
METADATA_TIE_CASES = {("A", "tags"), ("B", "tags"), ("B", "time")}


def metadata_inputs(g, tag):
    """Regenerate stream A / B of tests/golden/metadata.npz: (columns dict, labels, n, k)."""
    from mused_amd import synth

    n, k, seed, integer_time = (int(x) for x in g[f"{tag}_meta"])
    cols, labels = synth.metadata_stream(n, seed, integer_time=bool(integer_time))
    return cols, labels, n, k


def metadata_reference_adjacency(g, tag, t, n):
    return np.unpackbits(g[f"{tag}_{t}_bits"], axis=1)[:, :n]


def assert_valid_topk(A, valid, S, kk):
    """A (0/1, window coordinates) is A valid answer to "the kk smallest scores of every row of S, self removed":
    everything strictly below the kk-th smallest score is selected, nothing above it is, and the count is right.
    This is all the reference defines when scores tie (its argsort is unstable)."""
    kk = min(kk, len(valid))
    sub = np.asarray(A)[np.ix_(valid, valid)].astype(bool)
    other = ~np.eye(len(valid), dtype=bool)
    thr = np.partition(S, kk - 1, axis=1)[:, kk - 1][:, None]
    assert not ((S < thr) & ~sub & other).any()      # nothing closer was left out
    assert not (sub & (S > thr)).any()               # nothing farther was taken
    self_in = (np.diag(S)[:, None] < thr).ravel() | ((np.diag(S)[:, None] == thr).ravel() & (sub.sum(1) == kk - 1))
    assert np.array_equal(sub.sum(1) + self_in, np.full(len(valid), kk))
    rest = np.setdiff1d(np.arange(len(A)), valid)
    assert not np.asarray(A)[rest].any() and not np.asarray(A)[:, rest].any()


COSINE_DENSE_CASES = [("gauss", 500, 64, 0, 10), ("gauss", 500, 64, 0, 50), ("blob", 700, 96, 1, 10), ("blob", 700, 96, 1, 50)]


def cosine_dense_inputs(g, tag, n, d, seed, k):
    from mused_amd import synth

    X = synth.gauss_stream(n, d, seed)[0] if tag == "gauss" else synth.blob_stream(n, d, seed, n_centres=4)[0]
    assert synth.array_digest(X) == str(g[f"dense_{tag}_k{k}_digest"])
    return X


@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()
