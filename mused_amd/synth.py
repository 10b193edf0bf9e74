"""Seeded synthetic feature streams.

NumPy only; used by bench.py, the tests and the golden-fixture generator so
that inputs are regenerated from a seed instead of being stored.

  gauss : iid N(0, 1) rows (the BASELINE.json wording).
  blob  : `n_centres` centres ~ N(0, sep^2 I_d); label uniform; row = centre +
          N(0, I).  Labels give `n_clusters` per window (main.py:41).
  fd    : A = S D U + N / zeta, the Frequent-Directions paper generator implied
          by the file name at data_loader.py:191 (m = signal rank, zeta = SNR).

All streams are float32-valued (storage dtype of the device path); callers
promote to float64 where the reference does.
"""
from __future__ import annotations

import hashlib

import numpy as np


def gauss_stream(n: int, d: int, seed: int = 0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    # structure-less stream: labels only fix n_clusters per window (main.py:41)
    labels = np.random.default_rng(seed + 1000003).integers(0, 4, size=n).astype(np.int64)
    return X, labels


def blob_stream(n: int, d: int, seed: int = 0, n_centres: int = 8, sep: float = 2.0):
    rng = np.random.default_rng(seed)
    centres = (sep * rng.standard_normal((n_centres, d))).astype(np.float32)
    labels = rng.integers(0, n_centres, size=n)
    X = centres[labels] + rng.standard_normal((n, d), dtype=np.float32)
    return X.astype(np.float32), labels.astype(np.int64)


def fd_stream(n: int, d: int, seed: int = 0, m: int = 10, zeta: float = 10.0):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((n, m))
    D = np.diag(1.0 - np.arange(m) / m)
    U, _ = np.linalg.qr(rng.standard_normal((d, m)))
    A = S @ D @ U.T + rng.standard_normal((n, d)) / zeta
    return A.astype(np.float32), np.zeros(n, dtype=np.int64)


def two_modality_blob_stream(n: int, d_each: int, seed: int = 0, n_centres: int = 8, sep: float = 2.0):
    """BASELINE config 4: two d_each-dim modalities sharing one label sequence."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, n_centres, size=n)
    mods = []
    for _ in range(2):
        centres = (sep * rng.standard_normal((n_centres, d_each))).astype(np.float32)
        mods.append((centres[labels] + rng.standard_normal((n, d_each), dtype=np.float32)).astype(np.float32))
    return mods, labels.astype(np.int64)


def text_stream(n: int, seed: int = 0, vocab: int = 120, topics: int = 4, blank_rate: float = 0.04):
    """Synthetic ('title', 'description') string records for the reference's "text" modality
    (matrix_operations.py:91-110): every document draws its words from a topic-specific distribution over a small
    vocabulary (so that any two documents share words: no exact-zero cosine ties), a few rows have a blank title,
    a blank description, or both (both blank = invalid row, :96).  Returns ((n, 2) array of str, topic labels)."""
    rng = np.random.default_rng([seed, 0x7E57])
    words = np.array([f"w{i:03d}" for i in range(vocab)])
    base = rng.dirichlet(np.full(vocab, 0.6))
    dist = np.array([0.35 * base + 0.65 * rng.dirichlet(np.full(vocab, 0.08)) for _ in range(topics)])
    labels = rng.integers(0, topics, size=n)
    data = np.empty((n, 2), dtype=object)
    for i in range(n):
        p = dist[labels[i]]
        data[i, 0] = " ".join(rng.choice(words, size=int(rng.integers(4, 9)), p=p))
        data[i, 1] = " ".join(rng.choice(words, size=int(rng.integers(25, 50)), p=p))
        u = rng.random()
        if u < blank_rate:
            data[i, 0] = ""
        elif u < 2 * blank_rate:
            data[i, 1] = ""
        elif u < 2.5 * blank_rate:
            data[i, 0] = data[i, 1] = ""
    return data.astype(str), labels.astype(np.int64)


def metadata_stream(n: int, seed: int = 0, events: int = 5, users: int = 24, vocab: int = 60, missing: float = 0.05,
                    integer_time: bool = False):
    """Synthetic SED2012-style metadata columns, in the layout data_loader.py:87-99 hands to the reference:
        location (n, 2) float64  latitude, longitude in degrees, NaN where the photo has no geotag
        time     (n, 2) float64  datetaken, dateupload in seconds, 0.0 = missing stamp (matrix_operations.py:36)
        username (n, 1) str      '' = missing
        tags     (n, 1) object   list of str per row ([] = no tags)
    Every row belongs to one of `events` events: a place, a time span, a crowd of users and a tag distribution.
    Time stamps carry random fractions of a second unless integer_time (then equal differences -- ties -- are common).
    Returns (dict of arrays, event labels)."""
    rng = np.random.default_rng([seed, 0x3E7A])
    labels = rng.integers(0, events, size=n)
    centre = np.stack([rng.uniform(35.0, 60.0, events), rng.uniform(-10.0, 30.0, events)], axis=1)
    loc = centre[labels] + rng.normal(0.0, 0.05, size=(n, 2))
    loc[rng.random(n) < missing] = np.nan
    t0 = rng.uniform(1.2e9, 1.3e9, events)
    taken = t0[labels] + rng.uniform(0.0, 3 * 86400.0, n)
    upload = taken + rng.exponential(5 * 86400.0, n)
    time = np.stack([taken, upload], axis=1)
    if integer_time:
        time = np.floor(time / 3600.0) * 3600.0
    time[rng.random(n) < missing, 0] = 0.0
    time[rng.random(n) < missing, 1] = 0.0
    crowd = [rng.choice(users, size=max(3, users // events), replace=False) for _ in range(events)]
    user = np.array([[f"user{int(rng.choice(crowd[e])):03d}"] for e in labels], dtype=object)
    user[rng.random(n) < missing, 0] = ""
    words = [f"tag{i:02d}" for i in range(vocab)]
    dist = [0.3 * np.full(vocab, 1.0 / vocab) + 0.7 * rng.dirichlet(np.full(vocab, 0.1)) for _ in range(events)]
    tags = np.empty((n, 1), dtype=object)
    for i in range(n):
        m = int(rng.integers(0, 8))
        if rng.random() < missing:
            m = 0
        tags[i, 0] = [words[j] for j in rng.choice(vocab, size=m, p=dist[labels[i]])]  # repeats allowed: set() dedupes
    return {"location": loc, "time": time, "username": user.astype(str), "tags": tags}, labels.astype(np.int64)


STREAMS = {"gauss": gauss_stream, "blob": blob_stream, "fd": fd_stream}


def make_stream(kind: str, n: int, d: int, seed: int = 0, **kw):
    return STREAMS[kind](n, d, seed, **kw)


def array_digest(a: np.ndarray) -> str:
    """SHA-256 of dtype, shape and C-order bytes (used to pin regenerated inputs)."""
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    h.update(str(a.dtype).encode())
    h.update(str(a.shape).encode())
    h.update(a.tobytes())
    return h.hexdigest()


def stream_window_mods(window_index: int, W: int, dims, seed: int = 0, n_centres: int = 8, sep: float = 2.0):
    """Window `window_index` of an unbounded MULTI-modality blob stream (BASELINE config 4: dims = (512, 512)): one label
    sequence, one set of centres per modality; generated independently per window like `stream_window`.
    Returns (rows float32 (W, sum(dims)) -- the modalities side by side --, labels int64 (W,))."""
    base = np.random.default_rng([seed, 0x5EED, len(dims)])
    rng = np.random.default_rng([seed, 1 + window_index, len(dims)])
    labels = rng.integers(0, n_centres, size=W)
    parts = []
    for dm in dims:
        centres = (sep * base.standard_normal((n_centres, dm))).astype(np.float32)
        parts.append((centres[labels] + rng.standard_normal((W, dm), dtype=np.float32)).astype(np.float32))
    return np.concatenate(parts, axis=1), labels.astype(np.int64)


def stream_window(kind: str, window_index: int, W: int, d: int, seed: int = 0, n_centres: int = 8, sep: float = 2.0):
    """Window `window_index` of an unbounded stream, generated independently of the others (so ranks
    can materialise only their own windows).  Centres / mixing matrices depend on `seed` only; rows on
    (seed, window_index).  Returns (rows float32 (W, d), labels int64 (W,))."""
    base = np.random.default_rng([seed, 0x5EED])
    rng = np.random.default_rng([seed, 1 + window_index])
    if kind == "gauss":
        return rng.standard_normal((W, d), dtype=np.float32), rng.integers(0, 4, size=W).astype(np.int64)
    if kind == "blob":
        centres = (sep * base.standard_normal((n_centres, d))).astype(np.float32)
        labels = rng.integers(0, n_centres, size=W)
        X = centres[labels] + rng.standard_normal((W, d), dtype=np.float32)
        return X.astype(np.float32), labels.astype(np.int64)
    if kind == "fd":
        m, zeta = 10, 10.0
        U, _ = np.linalg.qr(base.standard_normal((d, m)))
        D = np.diag(1.0 - np.arange(m) / m)
        A = rng.standard_normal((W, m)) @ D @ U.T + rng.standard_normal((W, d)) / zeta
        return A.astype(np.float32), np.zeros(W, dtype=np.int64)
    raise ValueError(kind)
