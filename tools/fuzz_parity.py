"""Randomised parity sweep on the GPU: SWFD (device vs oracle/swfd_oracle.py) and kNN adjacency (device vs
oracle/mo_oracle.py) over random small shapes, batchings, dtypes and degenerate rows.  Prints one line per case and a
summary; exit code 1 on the first mismatch (the failing case is reproducible from its seed).

    python tools/fuzz_parity.py [--cases 60] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mused_amd import matrix_operations as mo
from mused_amd.swfd import SeqBasedSWFD as Dev
from oracle import mo_oracle as omo
from oracle.swfd_oracle import SeqBasedSWFD as Ora


def compare(dev, ora, tag):
    Bd, sd, ld, dd = dev.get()
    Bo, so, lo, do = ora.get()
    assert ld == lo, f"{tag}: level {ld} vs {lo}"
    s0 = max(so[0], 1e-300)
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, "\n dev sigma", sd, "\n ora sigma", so, "\n delta", dd, do, "level", ld, lo, flush=True)
    np.testing.assert_allclose(sd, so, rtol=0, atol=1e-8 * s0, err_msg=tag)
    np.testing.assert_allclose(Bd.T @ Bd, Bo.T @ Bo, rtol=0, atol=1e-8 * s0 * s0, err_msg=tag)
    assert abs(dd - do) <= 1e-8 * s0 * s0, tag


def swfd_case(rng, i):
    ell = int(rng.integers(2, 25))
    N = int(rng.integers(max(ell + 1, 8), 420))
    d = int(rng.integers(2, 90))
    kind = rng.choice(["gauss", "lowrank", "heavy", "binary", "zeros", "const"])
    n = int(rng.integers(N // 2, 3 * N + 50))
    if kind == "gauss":
        X = rng.standard_normal((n, d)) * rng.uniform(0.5, 20)
    elif kind == "lowrank":
        r = int(rng.integers(1, min(d, 6) + 1))
        X = rng.standard_normal((n, r)) @ rng.standard_normal((r, d)) + 1e-3 * rng.standard_normal((n, d))
    elif kind == "heavy":
        X = rng.standard_normal((n, d))
        X[rng.random(n) < 0.1] *= 40.0
    elif kind == "binary":
        X = (rng.random((n, d)) < 0.15).astype(np.int64)
    elif kind == "zeros":
        X = rng.standard_normal((n, d))
        X[rng.random(n) < 0.3] = 0.0
    else:
        X = np.tile(rng.standard_normal((1, d)), (n, 1))  # every row the same: rank one, exact duplicates
    Xf = X.astype(np.float64)
    R = max(float((Xf ** 2).sum(1).max()), 1.0) * float(rng.choice([1.0, 1.0, 3.7]))
    dev, ora = Dev(N=N, R=R, d=d, sketch_dim=ell), Ora(N=N, R=R, d=d, sketch_dim=ell)
    assert dev.L == ora.L
    t = 0
    checks = 0
    while t < n:
        step = int(rng.choice([1, 2, ell, ell + 1, N - 1, N, N + 1, int(rng.integers(1, 2 * N))]))
        step = max(1, min(step, n - t))
        blk = X[t:t + step]
        if rng.random() < 0.3:
            dev.fit(torch.from_numpy(np.ascontiguousarray(blk)).cuda())
        else:
            dev.fit(blk)
        ora.fit(blk)
        t += step
        if rng.random() < 0.5 or t == n:
            compare(dev, ora, f"swfd case {i} kind={kind} N={N} d={d} l={ell} t={t}")
            checks += 1
    dev.close()
    return f"swfd {kind:8s} N={N:3d} d={d:2d} l={ell:2d} rows={n:4d} L={ora.L} checks={checks}"


def knn_case(rng, i):
    n = int(rng.integers(2, 700))
    d = int(rng.integers(1, 48))
    kind = rng.choice(["gauss", "dups", "lattice", "zeros", "nonfinite"])
    X = rng.standard_normal((n, d))
    if kind == "dups":
        X[rng.integers(0, n, n // 3)] = X[rng.integers(0, n, n // 3)]
    elif kind == "lattice":
        X = rng.integers(-2, 3, size=(n, d)).astype(np.float64)
    elif kind == "zeros":
        X[rng.random(n) < 0.2] = 0.0
    elif kind == "nonfinite":
        X[rng.random(n) < 0.1, 0] = rng.choice([np.nan, np.inf, -np.inf])
    if rng.random() < 0.5:
        X = X.astype(np.float32)
    t = str(rng.choice(["", "cosine"]))
    nv = int(np.isfinite(X).all(axis=1).sum())
    k = int(rng.integers(0, max(nv, 1) + 1)) if rng.random() < 0.3 else int(rng.integers(0, min(nv, 60) + 1))
    try:
        ref = omo.create_adjacency_matrix(X.astype(np.float64), t, k)
        err = None
    except ValueError as e:
        ref, err = None, e
    if err is not None:
        try:
            mo.create_adjacency_matrix(X, t, k)
        except ValueError:
            return f"knn  {kind:9s} n={n:3d} d={d:2d} k={k:3d} type={t!r:8s} both raise"
        raise AssertionError(f"knn case {i}: oracle raised {err!r}, device did not")
    got = mo.create_adjacency_matrix(X, t, k)
    note = "ok"
    if not np.array_equal(got, ref):
        # Rows may differ only where the choice hangs on scores that are equal in exact arithmetic (duplicate rows,
        # lattice data): the host BLAS gives such pairs scores that differ in the last bits depending on which
        # micro-kernel handled the column (so does the reference), the device gives them equal scores and takes the
        # smaller row.  Everything that differs must sit within rounding of the row's k-th smallest score.
        valid = np.where(np.isfinite(X).all(axis=1))[0]
        Xv = X[valid].astype(np.float64)
        S = omo.cosine_scores(Xv) if t == "cosine" else omo.sq_euclidean_scores(Xv)
        kk = min(k + 1, len(valid)) if t == "cosine" else max(1, k)
        thr = np.partition(S, kk - 1, axis=1)[:, kk - 1][:, None]
        diff = (got != ref)[np.ix_(valid, valid)]
        scale = np.maximum(np.abs(thr), np.abs(S).max() * 1e-3)
        assert (np.abs(S - thr)[diff] <= 1e-12 * np.broadcast_to(scale, S.shape)[diff]).all(), \
            f"knn case {i} kind={kind} n={n} d={d} k={k} type={t!r} dtype={X.dtype}: {int(diff.sum())} entries differ beyond ties"
        # (degrees may differ by one in such rows: whether the row itself is among the tied picks decides if one of
        #  the k + 1 selected columns is dropped as the diagonal)
        note = f"equal up to {int(diff.sum())} tie entries"
    return f"knn  {kind:9s} n={n:3d} d={d:2d} k={k:3d} type={t!r:8s} dtype={X.dtype} {note}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=-1, help="run this case index only")
    a = ap.parse_args()
    t0 = time.time()
    np.set_printoptions(linewidth=200, precision=10)
    for i in (range(a.cases) if a.only < 0 else [a.only]):
        rng = np.random.default_rng([a.seed, i])
        fn = swfd_case if i % 2 == 0 else knn_case
        print(f"[{i:3d} seed=({a.seed},{i})]", fn(rng, i), f"({time.time() - t0:.0f}s)", flush=True)
    print(f"all {a.cases} cases passed")


if __name__ == "__main__":
    main()
