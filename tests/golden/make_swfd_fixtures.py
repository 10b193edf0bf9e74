#!/usr/bin/env python3
"""Fixtures of the SWFDMC approach in the reference's wiring (main.py:58-76: SeqBasedSWFD over the rows of the fused
W x W adjacency, d = W) at the HEADLINE window size W = 10,000, one and two modalities.

These come from THIS REPO'S CPU oracle (oracle/mo_oracle.py pipeline + oracle/swfd_oracle.py) -- the reference's `swfd`
submodule is absent (PARITY UNPINNED, see oracle/swfd_oracle.py) -- computed here because the pure-Python sketch needs
minutes per window, too long for the GPU test run.  Inputs are regenerated from seeds by the tests (digest checked).

    python tests/golden/make_swfd_fixtures.py            # ~10 min on 8 cores
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mused_amd import synth  # noqa: E402
from oracle import mo_oracle as omo  # noqa: E402
from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD  # noqa: E402


def case(tag, W, dims, ell, k, seed, n_windows):
    mods, labels = [], None
    for t in range(n_windows):
        if len(dims) == 1:
            X, lab = synth.stream_window("blob", t, W, dims[0], seed)
        else:
            X, lab = synth.stream_window_mods(t, W, dims, seed)
        mods.append(X)
        labels = lab if labels is None else np.concatenate([labels, lab])
    X = np.concatenate(mods).astype(np.float64)
    parts, c0 = [], 0
    for dm in dims:
        parts.append(np.ascontiguousarray(X[:, c0 : c0 + dm]))
        c0 += dm
    trace = []
    t0 = time.time()
    out = omo.process_streaming_data(parts, [""] * len(parts), W, ell, k, seed, "SWFDMC", labels, swfd_cls=OraSWFD, trace=trace)
    print(tag, "oracle pipeline", round(time.time() - t0, 1), "s", flush=True)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", tag + ".npz"),
        meta=np.array([W, ell, k, seed, n_windows] + list(dims)),
        input_digest=np.array([synth.array_digest(p.astype(np.float32)) for p in parts]),
        sigma=np.array([tr["sigma"] for tr in trace]),
        raw=np.array([tr["raw"] for tr in trace]).astype(np.int16),
        all_clusters=np.asarray(out).astype(np.int16),
        labels_sha=np.array(hashlib.sha256(np.asarray(out).astype(np.int64).tobytes()).hexdigest()),
    )


def case_sketch_checks(tag, N, d, ell, steps, seed=0):
    """SeqBasedSWFD of the specification over a ragged feature stream: after every block the singular values, the level, the
    final shrink and a sampled 64 x 64 block of B^T B (the device test compares all of them).  Config 3's orders (l = 256:
    rotations of order 512, queries of order 768 / 1024) with a reduced window so that epoch ends are crossed in minutes."""
    rows = sum(steps)
    X, _ = synth.stream_window("blob", 0, rows, d, seed)
    X64 = X.astype(np.float64)
    R = float((X64 ** 2).sum(1).max())
    ora = OraSWFD(N=N, R=R, d=d, sketch_dim=ell)
    idx = np.linspace(0, d - 1, 64).astype(np.int64)
    sig, lvl, dlt, blk = [], [], [], []
    t, t0 = 0, time.time()
    for st in steps:
        ora.fit(X64[t:t + st])
        t += st
        B, s, l, dd = ora.get()
        sig.append(s)
        lvl.append(l)
        dlt.append(dd)
        blk.append((B[:, idx].T @ B[:, idx]))
    print(tag, "oracle sketch", round(time.time() - t0, 1), "s", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", tag + ".npz"), meta=np.array([N, d, ell, seed] + list(steps)),
                        input_digest=np.array(synth.array_digest(X)), R=np.array(R), sigma=np.array(sig), level=np.array(lvl),
                        delta=np.array(dlt), gram_idx=idx, gram_block=np.array(blk))


def case_headline_lanes(tag, W, d, ell, n_lanes, steps, seed=0):
    """The configuration bench.py times (N = W = 10,000, d = 1024, l = 128): the specification's sketch over the first sum(steps)
    rows of `n_lanes` different windows of the benchmark stream (one sketch each, R from window 0) -- singular values, level, final
    shrink and a sampled 64 x 64 block of B^T B after every step."""
    rows = sum(steps)
    R = float((synth.stream_window("blob", 0, W, d, seed)[0].astype(np.float64) ** 2).sum(1).max())
    idx = np.linspace(0, d - 1, 64).astype(np.int64)
    sig, lvl, dlt, blk, dig = [], [], [], [], []
    t0 = time.time()
    for b in range(n_lanes):
        X = synth.stream_window("blob", b, W, d, seed)[0][:rows]
        dig.append(synth.array_digest(X))
        ora = OraSWFD(N=W, R=R, d=d, sketch_dim=ell)
        t = 0
        s_b, l_b, d_b, g_b = [], [], [], []
        for st in steps:
            ora.fit(X[t:t + st].astype(np.float64))
            t += st
            B, s, l, dd = ora.get()
            s_b.append(s); l_b.append(l); d_b.append(dd); g_b.append(B[:, idx].T @ B[:, idx])
        sig.append(s_b); lvl.append(l_b); dlt.append(d_b); blk.append(g_b)
    print(tag, "oracle sketches", round(time.time() - t0, 1), "s", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", tag + ".npz"), meta=np.array([W, d, ell, seed, n_lanes] + list(steps)),
                        input_digest=np.array(dig), R=np.array(R), levels=np.array(ora.L), sigma=np.array(sig), level=np.array(lvl),
                        delta=np.array(dlt), gram_idx=idx, gram_block=np.array(blk))


if __name__ == "__main__":
    only = sys.argv[1:]  # optional: tags to (re)generate
    cases = [
        ("swfdmc_w10k_m1", 10000, (64,), 128, 50, 0, 1),
        ("swfdmc_w10k_m2", 10000, (32, 32), 128, 50, 0, 1),
        # three windows: crosses the AUX -> MAIN swap at both epoch starts, the expiry of the first window's snapshots and
        # two Hungarian matching steps at d = W = 10,000 (round 4; ~30 CPU-minutes)
        ("swfdmc_w10k_m1_3win", 10000, (64,), 128, 50, 0, 3),
    ]
    # the reference's own default parameters (main.py:305-313: window_size 2000, reduced_dim 50, k_basis 50), three windows
    cases.append(("swfdmc_refdef_3win", 2000, (256,), 50, 50, 0, 3))
    for c in cases:
        if not only or c[0] in only:
            case(*c)
    if not only or "swfd_headline_lanes" in only:
        case_headline_lanes("swfd_headline_lanes", 10000, 1024, 128, 3, (640, 512))
    if not only or "swfd_c3shape" in only:
        case_sketch_checks("swfd_c3shape", 10000, 4096, 256, (640,))
    if not only or "swfd_c3orders" in only:
        case_sketch_checks("swfd_c3orders", 1024, 4096, 256, (700, 324, 1, 999, 476))
