#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own modules.

Run only in the build container (needs /root/reference, read-only).  Imports
`/root/reference/matrix_operations.py` and `/root/reference/main.py` with empty
stub modules for the absent third-party imports that the hot path never
touches (`hdbscan`, `ot`, `incdbscan`, and the un-vendored `swfd` submodule),
feeds them seeded synthetic streams from `mused_amd.synth`, and stores ONLY
data (hashes, singular values, sampled embedding rows, labels) under
tests/golden/*.npz.  No reference source text is copied.

    python tests/golden/make_golden.py            # all small cases
    python tests/golden/make_golden.py --big      # also the W=10000, d=1024 window
"""
import argparse
import contextlib
import hashlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
for _name, _attrs in (
    ("hdbscan", []),
    ("ot", []),
    ("incdbscan", ["IncrementalDBSCAN"]),
    ("swfd", ["SeqBasedSWFD"]),
):
    _m = types.ModuleType(_name)
    for _a in _attrs:
        setattr(_m, _a, None)
    sys.modules[_name] = _m

import matrix_operations as ref_mo  # noqa: E402  (the reference module)
import main as ref_main  # noqa: E402
import metrics_evaluation as ref_me  # noqa: E402
from sklearn.decomposition import TruncatedSVD  # noqa: E402

from mused_amd import synth  # noqa: E402


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def nbr_hash(A: np.ndarray) -> str:
    """SHA-256 over the row-major list of (row, sorted neighbour columns)."""
    h = hashlib.sha256()
    for i in range(A.shape[0]):
        cols = np.flatnonzero(A[i]).astype(np.int32)
        h.update(np.int32(i).tobytes())
        h.update(np.int32(len(cols)).tobytes())
        h.update(cols.tobytes())
    return h.hexdigest()


def window_record(mods, types_, k, ell, seed, n_clusters, sample_rows=8):
    adjs = [quiet(ref_mo.create_adjacency_matrix, m.astype(np.float64), t, k) for m, t in zip(mods, types_)]
    fused = ref_mo.fuse_matrices(adjs)
    R = float(np.max(np.linalg.norm(fused, axis=1) ** 2))  # main.py:61
    emb = ref_mo.perform_svd_reduction(fused, ell, seed)
    svd = TruncatedSVD(n_components=min(ell, fused.shape[1] - 1), random_state=seed)
    emb2 = svd.fit_transform(fused)
    assert np.array_equal(emb, emb2)
    labels = quiet(ref_mo.perform_clustering, emb, n_clusters, seed)
    rows = np.linspace(0, emb.shape[0] - 1, sample_rows).astype(np.int64)
    deg = fused.sum(axis=1)
    return dict(
        adj_hash=np.array([nbr_hash(a) for a in adjs]),
        fused_hash=np.array(nbr_hash(fused)),
        fused_dtype=np.array(str(fused.dtype)),
        deg_min=np.array(deg.min()),
        deg_max=np.array(deg.max()),
        R=np.array(R),
        sigma=svd.singular_values_.copy(),
        emb_rows=rows,
        emb_sample=emb[rows].copy(),
        emb_fro=np.array(np.linalg.norm(emb)),
        emb_abs_colsum=np.abs(emb).sum(axis=0),
        kmeans_labels=labels.astype(np.int32),
    )


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def case_cosine():
    """a2: the cosine kernel of the reference's "text" branch (matrix_operations.py:91-110).
    (1) the branch itself on synthetic ('title', 'description') strings, through the reference's own function;
    (2) its arithmetic -- sklearn `cosine_similarity` + `np.argsort(-sim, axis=1)[:, :k + 1]` (:106-108) and the
        write loop (:123-130) -- on dense feature rows, which is what modality_type="cosine" computes here."""
    from sklearn.metrics.pairwise import cosine_similarity

    out = {}
    n, k, seed = 600, 20, 0
    data, labels = synth.text_stream(n, seed)
    A = quiet(ref_mo.create_adjacency_matrix, data, "text", k)
    out["text_meta"] = np.array([n, k, seed])
    out["text_digest"] = np.array(hashlib.sha256("\x1f".join(data.ravel()).encode()).hexdigest())
    out["text_adj_hash"] = np.array(nbr_hash(A))
    out["text_deg"] = A.sum(axis=1).astype(np.int32)
    # all-blank input: no edges (:109-110)
    out["blank_A"] = quiet(ref_mo.create_adjacency_matrix, np.array([["", ""]] * 4), "text", 2).astype(np.uint8)
    for tag, (X, _) in (("gauss", synth.gauss_stream(500, 64, 0)), ("blob", synth.blob_stream(700, 96, 1, n_centres=4))):
        for kk in (10, 50):
            X64 = X.astype(np.float64)
            sim = cosine_similarity(X64)
            idx = np.argsort(-sim, axis=1)[:, : kk + 1]
            Ad = np.zeros((len(X), len(X)))
            for i, row in enumerate(idx):
                for j in row:
                    if i != j:
                        Ad[i, j] = 1
            out[f"dense_{tag}_k{kk}_hash"] = np.array(nbr_hash(Ad))
            out[f"dense_{tag}_k{kk}_digest"] = np.array(synth.array_digest(X))
    save("cosine", **out)


def case_windows(tag, kind, n, d, W, ell, k, seed, two_mod=False, **kw):
    """Per-window records for the first few windows of a stream."""
    if two_mod:
        mods, labels = synth.two_modality_blob_stream(n, d, seed, **kw)
        types_ = ["", ""]
    else:
        X, labels = synth.make_stream(kind, n, d, seed, **kw)
        mods, types_ = [X], [""]
    out = dict(
        meta=np.array([n, d, W, ell, k, seed]),
        kind=np.array(kind),
        n_centres=np.array(kw.get("n_centres", 8)),
        input_digest=np.array([synth.array_digest(m) for m in mods]),
    )
    for w in range(n // W):
        sl = slice(w * W, (w + 1) * W)
        rec = window_record([m[sl] for m in mods], types_, k, ell, seed, len(np.unique(labels[sl])))
        for key, v in rec.items():
            out[f"w{w}_{key}"] = v
    save(tag, **out)


def case_stream(tag, kind, n, d, W, ell, k, seed, approach="sSVDMC", two_mod=False, ratio=1, **kw):
    """Whole-run event labels through the reference's own window loop (main.py:13-130)."""
    if two_mod:
        mods, labels = synth.two_modality_blob_stream(n, d, seed, **kw)
        types_ = ["", ""]
    else:
        X, labels = synth.make_stream(kind, n, d, seed, **kw)
        mods, types_ = [X], [""]
    mods64 = [m.astype(np.float64) for m in mods]
    captured = {}

    def fake_metrics(results, subset_size, noise_rate, label_mode, sorting, reduced_dim, k_basis,
                     window_size, clusters, true_labels, t1, t0):
        captured["clusters"] = np.asarray(clusters).copy()
        captured["true"] = np.asarray(true_labels).copy()
        return results

    orig = ref_me.compute_all_metrics
    ref_me.compute_all_metrics = fake_metrics
    try:
        quiet(
            ref_main.process_streaming_data, {}, mods64, types_, W, ell, k, len(np.unique(labels)), seed,
            approach, labels, ratio, 0.0, "types", False, 1.5, 2,
        )
    finally:
        ref_me.compute_all_metrics = orig
    save(
        tag,
        meta=np.array([n, d, W, ell, k, seed]),
        kind=np.array(kind),
        input_digest=np.array([synth.array_digest(m) for m in mods]),
        all_clusters=captured["clusters"].astype(np.int64),
        labels_sha=np.array(hashlib.sha256(captured["clusters"].astype(np.int64).tobytes()).hexdigest()),
    )


def case_bench_stream(tag, kind, n_windows, W, d, ell, k, seed, dims=None):
    """Event labels of the reference's own window loop (main.py:13-130, approach sSVDMC) over the first `n_windows`
    windows of the benchmark stream (mused_amd.synth.stream_window): what bench.py's `labels_sha16` is compared with.
    Stored per window (the chain is sequential from window 0, so any prefix is a valid golden).  `dims`: several
    modalities side by side (mused_amd.synth.stream_window_mods, BASELINE config 4), each its own adjacency (main.py:45-56)."""
    if dims is None:
        wins = [synth.stream_window(kind, t, W, d, seed) for t in range(n_windows)]
    else:
        wins = [synth.stream_window_mods(t, W, dims, seed) for t in range(n_windows)]
    X = np.concatenate([w[0] for w in wins]).astype(np.float64)
    labels = np.concatenate([w[1] for w in wins])
    mods, c0 = [], 0
    for dm in (dims or (d,)):
        mods.append(np.ascontiguousarray(X[:, c0 : c0 + dm]))
        c0 += dm
    captured = {}

    def fake_metrics(results, subset_size, noise_rate, label_mode, sorting, reduced_dim, k_basis,
                     window_size, clusters, true_labels, t1, t0):
        captured["clusters"] = np.asarray(clusters).copy()
        return results

    orig = ref_me.compute_all_metrics
    ref_me.compute_all_metrics = fake_metrics
    try:
        quiet(ref_main.process_streaming_data, {}, mods, [""] * len(mods), W, ell, k, len(np.unique(labels)), seed, "sSVDMC",
              labels, 1, 0.0, "types", False, 1.5, 2)
    finally:
        ref_me.compute_all_metrics = orig
    allc = captured["clusters"].astype(np.int64).reshape(n_windows, W)
    cum = [hashlib.sha256(allc[: t + 1].tobytes()).hexdigest()[:16] for t in range(n_windows)]
    save(tag, meta=np.array([n_windows, W, d, ell, k, seed]), kind=np.array(kind),
         window_digest=np.array([synth.array_digest(w[0]) for w in wins]),
         labels=allc.astype(np.int8), cumulative_sha16=np.array(cum))


def case_metadata():
    """the metadata branches of create_adjacency_matrix (matrix_operations.py:22-89) on synthetic
    SED2012-style columns (mused_amd.synth.metadata_stream), through the reference's own function.  Stored per type: the
    adjacency bits (np.packbits).  Stream A (seed 0) has fractional time stamps; stream B (seed 1) whole hours, where the
    reference's unstable argsort decides between equal differences.  Plus one fused window (location OR username OR
    text) through fuse_matrices / perform_svd_reduction / perform_clustering, and a whole run of the reference's
    window loop over the (location, username) columns."""
    out = {}
    n, k = 300, 8
    for tag, seed, integer_time in (("A", 0, False), ("B", 1, True)):
        cols, labels = synth.metadata_stream(n, seed, integer_time=integer_time)
        for t in ("location", "time", "username", "tags"):
            A = quiet(ref_mo.create_adjacency_matrix, cols[t], t, k)
            out[f"{tag}_{t}_bits"] = np.packbits(A.astype(bool), axis=1)
            out[f"{tag}_{t}_hash"] = np.array(nbr_hash(A))
        out[f"{tag}_meta"] = np.array([n, k, seed, int(integer_time)])
    # degenerate inputs: nothing valid -> no edges; fewer valid rows than k
    out["none_location"] = quiet(ref_mo.create_adjacency_matrix, np.full((4, 2), np.nan), "location", 2).astype(np.uint8)
    out["none_time"] = quiet(ref_mo.create_adjacency_matrix, np.zeros((4, 2)), "time", 2).astype(np.uint8)
    out["none_user"] = quiet(ref_mo.create_adjacency_matrix, np.array([[""]] * 4), "username", 2).astype(np.uint8)
    few, _ = synth.metadata_stream(6, 2, missing=0.0)
    for t in ("location", "time", "username", "tags"):
        out[f"few_{t}"] = quiet(ref_mo.create_adjacency_matrix, few[t], t, 8).astype(np.uint8)
    # fused window + eigenstep + labels
    n, k, ell, seed = 400, 10, 12, 0
    cols, labels = synth.metadata_stream(n, 3)
    text, _ = synth.text_stream(n, 3)
    types_ = ["location", "username", "text"]
    mods = [cols["location"], cols["username"], text]
    adjs = [quiet(ref_mo.create_adjacency_matrix, m, t, k) for m, t in zip(mods, types_)]
    fused = ref_mo.fuse_matrices(adjs)
    svd = TruncatedSVD(n_components=min(ell, fused.shape[1] - 1), random_state=seed)
    emb = svd.fit_transform(fused)
    assert np.array_equal(emb, ref_mo.perform_svd_reduction(fused, ell, seed))
    km = quiet(ref_mo.perform_clustering, emb, len(np.unique(labels)), seed)
    out["win_meta"] = np.array([n, k, ell, seed, 3])
    out["win_adj_hash"] = np.array([nbr_hash(a) for a in adjs])
    out["win_fused_hash"] = np.array(nbr_hash(fused))
    out["win_R"] = np.array(float(np.max(np.linalg.norm(fused, axis=1) ** 2)))
    out["win_sigma"] = svd.singular_values_
    out["win_labels"] = km.astype(np.int32)
    # whole run: (location, username) through main.py's loop
    n, W, ell, k, seed = 1200, 300, 8, 8, 0
    cols, labels = synth.metadata_stream(n, 4)
    captured = {}

    def fake_metrics(results, subset_size, noise_rate, label_mode, sorting, reduced_dim, k_basis,
                     window_size, clusters, true_labels, t1, t0):
        captured["clusters"] = np.asarray(clusters).copy()
        return results

    orig = ref_me.compute_all_metrics
    ref_me.compute_all_metrics = fake_metrics
    try:
        quiet(ref_main.process_streaming_data, {}, [cols["location"], cols["username"]], ["location", "username"], W, ell,
              k, len(np.unique(labels)), seed, "sSVDMC", labels, 1, 0.0, "types", False, 1.5, 2)
    finally:
        ref_me.compute_all_metrics = orig
    out["run_meta"] = np.array([n, W, ell, k, seed, 4])
    out["run_clusters"] = captured["clusters"].astype(np.int64)
    save("metadata", **out)


def case_edges():
    """Small edge cases of create_adjacency_matrix / fuse / match_clusters."""
    rng = np.random.default_rng(7)
    out = {}
    # (a) non-finite rows are dropped from the kNN and get empty adjacency rows/cols
    X = rng.standard_normal((40, 6))
    X[3, 2] = np.nan
    X[17, 0] = np.inf
    A = quiet(ref_mo.create_adjacency_matrix, X, "", 5)
    out["nonfinite_X"] = X
    out["nonfinite_A"] = A.astype(np.uint8)
    # (b) k_basis = 1 -> only self -> empty adjacency; k_basis = 0 -> max(1, k) = 1
    X = rng.standard_normal((12, 3))
    out["k1_X"] = X
    out["k1_A"] = quiet(ref_mo.create_adjacency_matrix, X, "", 1).astype(np.uint8)
    out["k0_A"] = quiet(ref_mo.create_adjacency_matrix, X, "", 0).astype(np.uint8)
    # (c) k = n (every other row is a neighbour)
    out["kn_A"] = quiet(ref_mo.create_adjacency_matrix, X, "", 12).astype(np.uint8)
    # (d) fuse dtype rule: 1 modality float64 copy, >= 2 int64
    A1 = quiet(ref_mo.create_adjacency_matrix, X, "", 4)
    A2 = quiet(ref_mo.create_adjacency_matrix, X[:, ::-1] * np.array([1.0, 3.0, 0.2]), "", 3)
    F1 = ref_mo.fuse_matrices([A1])
    F2 = ref_mo.fuse_matrices([A1, A2])
    out["fuse1_dtype"] = np.array(str(F1.dtype))
    out["fuse2_dtype"] = np.array(str(F2.dtype))
    out["fuse_A1"] = A1.astype(np.uint8)
    out["fuse_A2"] = A2.astype(np.uint8)
    out["fuse2"] = F2.astype(np.uint8)
    # (e) demo-sized rsvd where n_components + 10 > n (main.py:318-324: window 8, dim 2, k 1 -> k=2 here to get edges)
    Xd = rng.standard_normal((8, 4))
    Ad = quiet(ref_mo.create_adjacency_matrix, Xd, "", 3)
    out["demo_X"] = Xd
    out["demo_A"] = Ad.astype(np.uint8)
    out["demo_emb"] = ref_mo.perform_svd_reduction(ref_mo.fuse_matrices([Ad]), 2, 0)
    # (f) match_clusters: feasible, infeasible, prev None
    prev = rng.integers(0, 4, size=200)
    perm = np.array([2, 0, 3, 1])
    new = perm[prev].copy()
    flip = rng.random(200) < 0.1
    new[flip] = rng.integers(0, 4, size=int(flip.sum()))
    out["match_prev"] = prev
    out["match_new"] = new
    out["match_out"] = np.asarray(quiet(ref_mo.match_clusters, prev, new, "hungarian", 3))
    new_inf = np.where(np.arange(200) < 2, 9, new)  # label 9 overlaps < min_overlap with every prev label
    out["match_new_inf"] = new_inf
    out["match_out_inf"] = np.asarray(quiet(ref_mo.match_clusters, prev, new_inf, "hungarian", 3))
    save("edges", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated case tags to (re)generate")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    if only:
        if "cosine" in only:
            case_cosine()
        if "metadata" in only:
            case_metadata()
        if "c1_stream_hop2_blob_s0" in only:   # hopping windows (step_window_ratio = 2, main.py:32)
            case_stream("c1_stream_hop2_blob_s0", "blob", 3000, 64, 500, 16, 50, 0, ratio=2, n_centres=4, sep=2.0)
            case_stream("c1_stream_hop4_gauss_s1", "gauss", 2000, 64, 400, 16, 30, 1, ratio=4)
        if "bench_c2_blob_s0" in only:   # bench.py default stream: 20 windows of BASELINE config 2
            case_bench_stream("bench_c2_blob_s0", "blob", 20, 10000, 1024, 128, 50, 0)
        if "bench_c3_blob_s0" in only:   # bench.py --workload c3: 8 windows of BASELINE config 3
            case_bench_stream("bench_c3_blob_s0", "blob", 8, 10000, 4096, 256, 50, 0)
        if "bench_c4_blob_s0" in only:   # bench.py --workload c4: 8 windows of BASELINE config 4 (two 512-d modalities)
            case_bench_stream("bench_c4_blob_s0", "blob", 8, 10000, 1024, 128, 50, 0, dims=(512, 512))
        if "refdef_blob_s0" in only:   # the reference's OWN default parameters (main.py:305-313): W = 2000, reduced_dim = 50, k = 50
            case_windows("refdef_blob_s0", "blob", 2000, 256, 2000, 50, 50, 0, n_centres=8, sep=2.0)
            case_bench_stream("bench_refdef_blob_s0", "blob", 10, 2000, 256, 50, 50, 0)
        if "c3_blob_s0" in only:   # BASELINE config 3 at its real shape
            case_windows("c3_blob_s0", "blob", 10000, 4096, 10000, 256, 50, 0, n_centres=8, sep=2.0)
        if "c4_twomod_s0" in only:  # BASELINE config 4 at its real shape (one of the 8 windows)
            case_windows("c4_twomod_s0", "blob2", 10000, 512, 10000, 128, 50, 0, two_mod=True, n_centres=8)
        return
    case_edges()
    case_cosine()
    case_metadata()
    # BASELINE config 1 shapes: n=5000, d=64, W=500, ell=16, k=50
    for seed in (0, 1):
        case_windows(f"c1_gauss_s{seed}", "gauss", 1500, 64, 500, 16, 50, seed)
        case_windows(f"c1_blob_s{seed}", "blob", 1500, 64, 500, 16, 50, seed, n_centres=4, sep=2.0)
    case_windows("c1_fd_s0", "fd", 1000, 64, 500, 16, 50, 0)
    case_windows("c4s_twomod_s0", "blob2", 1024, 32, 512, 16, 20, 0, two_mod=True, n_centres=4)
    for seed in (0, 1):
        case_stream(f"c1_stream_blob_s{seed}", "blob", 5000, 64, 500, 16, 50, seed, n_centres=4, sep=2.0)
    case_stream("c1_stream_gauss_s0", "gauss", 5000, 64, 500, 16, 50, 0)
    case_stream("c1_stream_hop2_blob_s0", "blob", 3000, 64, 500, 16, 50, 0, ratio=2, n_centres=4, sep=2.0)
    case_stream("c1_stream_hop4_gauss_s1", "gauss", 2000, 64, 400, 16, 30, 1, ratio=4)
    case_stream("c4s_stream_twomod_s0", "blob2", 2048, 32, 512, 16, 20, 0, two_mod=True, n_centres=4)
    # a mid-size window in the C2 aspect ratio that the CPU suite can afford
    case_windows("c2m_blob_s0", "blob", 2000, 256, 2000, 64, 50, 0, n_centres=8, sep=2.0)
    # the reference's own operating point (main.py:305-313: window_size 2000, reduced_dim 50, k_basis 50): one window record
    # and ten windows through its window loop (approach sSVDMC)
    case_windows("refdef_blob_s0", "blob", 2000, 256, 2000, 50, 50, 0, n_centres=8, sep=2.0)
    case_bench_stream("bench_refdef_blob_s0", "blob", 10, 2000, 256, 50, 50, 0)
    if args.big:
        case_windows("c2_blob_s0", "blob", 10000, 1024, 10000, 128, 50, 0, n_centres=8, sep=2.0)
        case_windows("c2_gauss_s0", "gauss", 10000, 1024, 10000, 128, 50, 0)
        case_windows("c3_blob_s0", "blob", 10000, 4096, 10000, 256, 50, 0, n_centres=8, sep=2.0)
        case_windows("c4_twomod_s0", "blob2", 10000, 512, 10000, 128, 50, 0, two_mod=True, n_centres=8)


if __name__ == "__main__":
    main()
