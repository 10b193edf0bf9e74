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


if __name__ == "__main__":
    only = sys.argv[1:]  # optional: tags to (re)generate
    cases = [
        ("swfdmc_w10k_m1", 10000, (64,), 128, 50, 0, 1),
        ("swfdmc_w10k_m2", 10000, (32, 32), 128, 50, 0, 1),
        # three windows: crosses the AUX -> MAIN swap at both epoch starts, the expiry of the first window's snapshots and
        # two Hungarian matching steps at d = W = 10,000 (round 4; ~30 CPU-minutes)
        ("swfdmc_w10k_m1_3win", 10000, (64,), 128, 50, 0, 3),
    ]
    for c in cases:
        if not only or c[0] in only:
            case(*c)
