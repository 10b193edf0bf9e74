#!/usr/bin/env python3
"""Headline benchmark: stream rows/s (+ p50 per-window latency) of the MI355X hot path on the
BASELINE.json config-2 workload: synthetic stream, d = 1024, l = 128, window W = N = 10,000, k = 50.

One STEP = one window of W rows, already resident in HBM, through the whole path:
    SeqBasedSWFD over the W feature rows (append + get)         [a5-a7]
    Euclidean kNN adjacency of the window (fp64 MFMA scores + exact selection), per modality   [a1]
    fusion / R                                                  [a3, a4]
    randomized-SVD eigenstep on the fused adjacency             [a8]
    k-means + Hungarian matching on the host (sklearn / SciPy)  [a10]  -> event labels
`value` = rows of all ranks / wall time of the K timed steps (max over ranks), inputs resident.

Stream layout (the same for every lane / rank count): rank r owns the K timed windows [r K, (r + 1) K) of the
seeded stream; inside a rank they are dealt as CONTIGUOUS blocks to B = --lanes "lanes" (fixed: it does not depend
on --steps), each preceded by its halo / warm-up windows (the windows just before its block).  The sketches of the
lanes advance in lock-step inside shared launches, in up to three groups on three HIP streams; a fourth,
high-priority stream runs adjacency -> eigenstep of the same windows and a pool of host workers the k-means /
matching.  When K is not a multiple of B the shorter lanes idle in the last lock-step (`config.padded_window_slots`
says how many slots): that time is part of `value`.  `single_lane` is what ONE stream consumed strictly in order gets
(no batching across windows): rows/s and p50 window latency.

    python bench.py                       # 1 GPU, K = 20, W = 5 (what the driver runs)
    python bench.py --gpus N              # starts N ranks itself (a child `python -m torch.distributed.run`, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W     # the driver's form: used as launched

Multi-GPU: every rank owns a contiguous block of windows of the stream (mused_amd/distributed.py):
weak scaling, no data-path collective; one all-gather of raw labels at the end.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# HIP maps streams onto a few hardware queues (4 by default) and streams that share one run in order.  This script
# keeps 4 streams busy per rank (3 sketch groups + the main path) next to the default stream and whatever the
# collective library opens: with 8 queues the result no longer depends on how many other streams exist (with 4, two idle
# extra streams cost a third of the throughput).  Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration `metric` is quoted on
    "c2": dict(W=10000, dims=(1024,), ell=128, k=50, lanes=20,
               name="synthetic d=1024 l=128 window=10000 k=50 (BASELINE config 2)"),
    # BASELINE.json configs[2]
    "c3": dict(W=10000, dims=(4096,), ell=256, k=50, lanes=8,
               name="synthetic d=4096 l=256 window=10000 k=50 (BASELINE config 3)"),
    # BASELINE.json configs[3]: two 512-d modalities -> two kNN adjacencies -> OR-fusion (reference semantics,
    # main.py:45-56); the feature-row sketch sees the 1024-d concatenated rows
    "c4": dict(W=10000, dims=(512, 512), ell=128, k=50, lanes=20,
               name="synthetic two modalities d=512+512 l=128 window=10000 k=50 (BASELINE config 4)"),
    # small plumbing case (configs[0] shapes) for quick checks
    "c1": dict(W=500, dims=(64,), ell=16, k=50, lanes=4, name="synthetic d=64 l=16 window=500 k=50 (BASELINE config 1)"),
    # the reference's OWN default parameters (/root/reference/main.py:305-313: window_size 2000, reduced_dim 50, k_basis 50) on a
    # 256-d synthetic stream: rotations of order 100, queries of order 150, eigenstep at r = 60 -- all on the direct solver
    "refdef": dict(W=2000, dims=(256,), ell=50, k=50, lanes=10,
                   name="synthetic d=256 l=50 window=2000 k=50 (the reference's default parameters, main.py:305-313)"),
    # the reference's OWN use of the sketch (main.py:58-76, approach SWFDMC): SeqBasedSWFD over the rows of the fused W x W
    # adjacency (d = W = 10,000, bit rows), R from the first window, sketch transposed to (W, l) -> k-means -> matching.
    # Round 4: contiguous blocks of windows on lock-step lanes, each preceded by its halo window (mused_amd.pipeline.SwfdmcLanes).
    "swfdmc": dict(W=10000, dims=(1024,), ell=128, k=50, lanes=12,
                   name="SWFDMC wiring of main.py:58-76: features d=1024 -> kNN adjacency -> sketch over its W=10000 bit rows, l=128"),
}
PRE_STREAM = 1 << 20  # window indices of warm-up windows that would precede window 0 of the stream
FP64_PEAK_TFLOPS = 78.6  # gfx950: the fp64 vector-FMA peak and the fp64 MFMA peak are the same number
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--kind", default="blob", choices=["blob", "gauss", "fd"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-swfd", action="store_true", help="diagnostic: skip the feature-row SWFD stage")
    ap.add_argument("--no-single-lane", action="store_true", help="skip the in-order single-lane measurement")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse the N-rank launch only: every rank joins the process group (gloo, CPU), one all-reduce, "
                         "rank 0 prints the world size -- no GPU is touched")
    ap.add_argument("--lanes", type=int, default=0,
                    help="contiguous blocks of the rank's windows whose sketches advance in lockstep inside the same "
                         "launches (0 = the workload's default; 1 = strictly one window at a time)")
    ap.add_argument("--sketch-groups", type=int, default=0,
                    help="independent groups of lanes, each on its own HIP stream / host thread (0 = auto: 3 groups "
                         "from 6 lanes, 2 from 4): their launches interleave on the GPU, so the Gram / rotate GEMMs of one "
                         "group overlap the Jacobi rounds of the others and partly filled workgroup rounds are shared.  More "
                         "than 3 groups + the main stream exceed the hardware queues HIP hands out and serialise.")
    return ap.parse_args()


def hip_event_ms(fn, stream, reps=1):
    """Average duration (ms) of fn() measured with HIP events recorded on `stream`
    (the stream the kernels are launched on)."""
    import torch

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def window_rows(cfg, kind, idx, seed):
    """(rows float32 (W, sum(dims)), labels) of window `idx` of the benchmark stream."""
    from mused_amd import synth

    if len(cfg["dims"]) == 1:
        return synth.stream_window(kind, idx, cfg["W"], cfg["dims"][0], seed)
    return synth.stream_window_mods(idx, cfg["W"], cfg["dims"], seed)


def split_mods(cfg, rows):
    """Column slices of a resident (W, sum(dims)) window: one per modality (views, no copy)."""
    out, c = [], 0
    for dm in cfg["dims"]:
        out.append(rows[:, c : c + dm])
        c += dm
    return out


def stage_profile(cfg, X, pipe, sketch, rows_grp):
    """Per-stage device times (ms, HIP events on the launch stream) for one resident window."""
    import ctypes as C  # noqa: F401

    import torch

    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    W, ell, k = cfg["W"], cfg["ell"], cfg["k"]
    eng = pipe.eng
    st = torch.cuda.current_stream()
    out = {}
    X0 = split_mods(cfg, X)[0]
    d0 = X0.shape[1]
    dt = _lib.F32 if X0.dtype == torch.float32 else _lib.F64
    f_scores = lambda: _lib.call("mused_pairwise_scores", ptr(X0), dt, W, d0, X0.stride(0), 0, ptr(eng.norms),
                                 ptr(eng.scores), stream_ptr())
    f_scores()
    out["scores_gemm_ms"] = hip_event_ms(f_scores, st, 3)   # classic path (score matrix), for comparison
    w = (W + 63) // 64
    mask = torch.empty((W, w), dtype=torch.int64, device="cuda")
    f_sel = lambda: _lib.call("mused_select_k_smallest", ptr(eng.scores), W, W, k, None, ptr(mask), w, stream_ptr())
    f_sel()
    out["select_ms"] = hip_event_ms(f_sel, st, 3)
    f_knn = lambda: eng.knn_adjacency(X0, k)
    f_knn()
    out["knn_fused_ms"] = hip_event_ms(f_knn, st, 3)   # the product path: similarity + selection, no score matrix
    adj = eng.knn_adjacency(X0, k)
    f_rsvd = lambda: eng.svd_reduce(adj, ell, pipe.seed, nnz_cap=W * k * len(cfg["dims"]))
    f_rsvd()
    out["rsvd_ms"] = hip_event_ms(f_rsvd, st, 2)
    if sketch is not None:
        f_app = lambda: sketch.fit_lanes(rows_grp[:, -1])
        out["swfd_append_ms_group0_alone"] = hip_event_ms(f_app, st, 1)
        f_get = lambda: sketch.get_device()
        out["swfd_query_ms_group0_alone"] = hip_event_ms(f_get, st, 1)
        out["swfd_levels"] = sketch.L
        out["swfd_lanes_group0"] = sketch.lanes
    return out


def cpu_baseline(cfg, kind, seed, with_swfd=True):
    """The CPU oracle (oracle/*.py, a port of the reference path pinned to its golden vectors) timed
    on this box's host cores over a bounded sample of the same workload.  BLAS / OpenMP pools are capped at
    16 threads (a one-GPU share of the host): left at one thread per visible core (256 here) the same sample
    runs ~10x slower, which would flatter the GPU.  The SWFD part is this repo's OWN specification of the absent
    `swfd` submodule (parity unpinned): `value_without_swfd` is the figure that rests on reference-pinned code only."""
    from oracle import mo_oracle as omo
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    threads = max(1, min(16, os.cpu_count() or 1))
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter, threads = None, os.cpu_count() or 1
    W, ell, k = cfg["W"], cfg["ell"], cfg["k"]
    X, labels = window_rows(cfg, kind, 0, seed)
    mods, c = [], 0
    for dm in cfg["dims"]:
        mods.append(X[:, c : c + dm].astype(np.float64))
        c += dm
    t0 = time.perf_counter()
    F = omo.fuse_matrices([omo.create_adjacency_matrix(m, "", k) for m in mods])
    omo.max_row_sq_norm(F)
    t1 = time.perf_counter()
    emb, _, _ = omo.randomized_svd_reduce(F, ell, seed)
    t2 = time.perf_counter()
    omo.perform_clustering(emb, len(np.unique(labels)), seed)
    t3 = time.perf_counter()
    swfd_rows = 0
    t_swfd_per_row = 0.0
    if with_swfd:
        X64 = X.astype(np.float64)
        R = float((X64**2).sum(1).max())
        sk = OraSWFD(N=W, R=R, d=X.shape[1], sketch_dim=ell)
        # blocks of l rows (one rotation of every level each; the steady-state cost per row is constant) until ~10 s of
        # CPU time are spent, at least 4 blocks
        ts = time.perf_counter()
        while swfd_rows < W and (swfd_rows < 4 * ell or time.perf_counter() - ts < 10.0):
            sk.fit(X64[swfd_rows:swfd_rows + ell])
            swfd_rows = min(W, swfd_rows + ell)
        sk.get()
        t_swfd_per_row = (time.perf_counter() - ts) / swfd_rows
    if limiter is not None:
        limiter.restore_original_limits()
    window_s = (t3 - t0) + t_swfd_per_row * W
    return {
        "value": W / window_s,
        "unit": "rows/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 window of {W} rows through oracle adjacency+fuse ({t1 - t0:.2f}s), eigenstep ({t2 - t1:.2f}s), "
                  f"k-means ({t3 - t2:.2f}s); "
                  + (f"SWFD oracle timed on {swfd_rows} rows ({t_swfd_per_row * 1e3:.2f} ms/row) and extrapolated to the window; "
                     if with_swfd else "no sketch stage (--no-swfd); ")
                  + f"BLAS/OpenMP pools capped at {threads} threads",
        "window_seconds": window_s,
        "value_without_swfd": W / (t3 - t0),
    }


def run_swfdmc(args, cfg):
    """`--workload swfdmc`: the reference's wiring of the sketch (main.py:58-76) at the headline window size.  Rank r owns
    the K timed windows [r K, (r + 1) K) of the seeded stream, dealt as contiguous blocks to B lanes whose sketches advance
    in lock-step (mused_amd.pipeline.SwfdmcLanes); every lane first sketches the window before its block (its halo; a window
    of empty rows at the very beginning of the stream) -- that lock-step is timed separately and charged like the feature-row
    workloads charge theirs.  --lanes 1 = one in-order stream."""
    import torch
    import torch.distributed as dist

    from mused_amd import distributed as mdist
    from mused_amd import matrix_operations as mo
    from mused_amd.pipeline import SwfdmcLanes

    rank, local_rank, world = rank_env(args)
    backend = os.environ.get("MUSED_DIST_BACKEND", "nccl")
    if "MUSED_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["MUSED_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    W, ell, k, d = cfg["W"], cfg["ell"], cfg["k"], cfg["dims"][0]
    K = args.steps
    B = max(1, min(args.lanes if args.lanes > 0 else cfg["lanes"], K))
    g0 = rank * K
    blocks = [mdist.block_partition(K, B, p) for p in range(B)]
    blk = max(b1 - b0 for b0, b1 in blocks)
    need = sorted({g0 + b0 - 1 + t for b0, b1 in blocks for t in range(0, 1 + (b1 - b0)) if g0 + b0 - 1 + t >= 0})
    host = {gi: window_rows(cfg, args.kind, gi, args.seed) for gi in need}
    dev = {gi: torch.from_numpy(host[gi][0]).cuda() for gi in need}            # resident before timing
    # R of main.py:61 comes from window 0 of the STREAM (every rank computes it from that window: no broadcast needed)
    x0 = torch.from_numpy(window_rows(cfg, args.kind, 0, args.seed)[0]).cuda()
    R = SwfdmcLanes.r_of_first_window([x0], W, k)
    del x0
    # lanes in G groups, each a SwfdmcLanes of its own on its own stream and host thread: the Gram / rotate GEMMs of one group
    # overlap the eigensolver chain and the memory-bound passes (bit expansion, scatter) of the other
    import threading

    G = args.sketch_groups if args.sketch_groups > 0 else (3 if B >= 9 else (2 if B >= 4 else 1))
    G = max(1, min(G, B))
    bounds = [mdist.block_partition(B, G, g) for g in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    groups = [SwfdmcLanes(W, ell, k, args.seed, l1 - l0, R, modality_types=[""], stream=streams[g], assume_finite=True)
              for g, (l0, l1) in enumerate(bounds)]
    lanes = groups[0]

    def group_step(g, t):
        l0, l1 = bounds[g]
        mods, labs, trig, want = [], [], [], []
        for b0, b1 in blocks[l0:l1]:
            idx = b0 - 1 + t
            own = t >= 1 and idx < b1
            gi = g0 + min(idx, b1 - 1)
            mods.append([dev[gi]] if gi >= 0 else None)
            labs.append(host[gi][1] if gi >= 0 else None)
            trig.append(gi)
            want.append(own)
        with torch.cuda.stream(streams[g]):
            groups[g].step(mods, labs, trig, want)

    def run_steps(t_lo, t_hi):
        def drive(g):
            torch.cuda.set_device(local_rank)  # the current device is per host thread
            for t in range(t_lo, t_hi):
                group_step(g, t)
            groups[g].pipe.flush()
            streams[g].synchronize()
        ths = [threading.Thread(target=drive, args=(g,)) for g in range(G)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    def bracket(fn):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        tm = torch.tensor([el], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    for gr in groups:
        gr.sk.profile(True)
    t_halo = bracket(lambda: run_steps(0, 1))            # the halo lock-step (also the warm-up: plans, graphs, caches)
    n_warm = [len(gr.pipe.latencies) for gr in groups]
    for gr in groups:
        gr.sk.profile(True)
    elapsed = bracket(lambda: run_steps(1, 1 + blk))
    raw = {}
    for gr in groups:
        gr.sk.check()
        raw.update({tr["trigger"]: tr["raw"] for tr in gr.pipe.trace})
    raw_local = np.array([raw[g0 + t] for t in range(K)], dtype=np.int64)
    raw_all = mdist.gather_raw_labels(raw_local, [K] * world, device=coll_dev)
    all_labels = mdist.replay_label_chain(raw_all, mo.match_clusters)
    if rank == 0:
        reads = [gr.sk.profile_read_direct() for gr in groups]
        direct = all(r[0] for r in reads)
        t_ms, n_launch, solved = sum(r[1] for r in reads), sum(r[2] for r in reads), sum(r[3] for r in reads)
        lat = np.concatenate([np.array(gr.pipe.latencies[n0:]) for gr, n0 in zip(groups, n_warm)])
        fl = 4.0 * 256 ** 3 / 3 + 2.0 * 256 * 256 * 128 + 20 * 512 * 255 * 5.0 + 128 * 4.0 * 255 * 6.0
        roof = None
        if direct and n_launch:
            us = 1e3 * t_ms / n_launch
            tfl = (solved / n_launch) * fl / (us * 1e-6) / 1e12
            roof = {"kernel": "trd_a .. trd_d kernels (direct eigensolver of the FD rotation, order 256; see the c2 line)", "bound": "mfma",
                    "bound_detail": "fp64 vector ALU + fp64 MFMA back-transformation, priced against the 78.6 TFLOP/s both share",
                    "achieved": tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP64_PEAK_TFLOPS, "traffic": None,
                    "launch_us": us, "launches_timed": n_launch, "matrices_solved_per_launch_avg": solved / n_launch}
        halo_share = t_halo * K / 100.0
        out = np.asarray(all_labels, dtype=np.int64)
        print(json.dumps({
            "metric": "stream rows/sec, SWFDMC wiring (main.py:58-76): kNN adjacency -> SWFD over its 10,000 bit rows -> labels",
            "value": world * K * W / (elapsed + halo_share), "unit": "rows/s", "n_gpus": world, "steps": K, "warmup": 1,
            "ms_per_step": 1e3 * (elapsed + halo_share) / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "value_excl_halo": world * K * W / elapsed,
            "halo": {"lock_step_s": t_halo, "share_charged_s": halo_share, "timed_s": elapsed,
                     "rule": "value = rows / (timed + halo lock-step x K / 100): every lane sketches one window it does not own "
                             "before its block; on a 100-window-per-GPU stream the K timed windows carry K / 100 of that lock-step"},
            "config": {"workload": cfg["name"], "stream": args.kind, "W": W, "d_features": d, "d_sketch": W, "l": ell, "k": k,
                       "swfd_levels": lanes.sk.L, "R": R, "lanes": B, "lane_groups": [l1 - l0 for l0, l1 in bounds],
                       "parallelism": f"windows sharded in contiguous blocks over {world} GPU(s) x {B} lock-step lane(s) per GPU",
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                       "collective_backend": (dist.get_backend() if world > 1 else None),
                       "labels_sha16": hashlib.sha256(out.tobytes()).hexdigest()[:16]},
            "p50_window_latency_ms": float(np.median(lat) * 1e3) if len(lat) else None,
            "roofline": roof,
            "cpu_baseline": None,
            "note": "parity of this wiring at W = 10,000, lanes against the sequential specification over three windows: "
                    "tests/test_gpu_headline_shapes.py (oracle fixtures, PARITY UNPINNED: the reference's swfd submodule is absent)",
        }))
    for gr in groups:
        gr.close()
    if world > 1:
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD process (never exec: this process may
    not be replaced once anything has touched the GPU, and the child's exit code is ours) -- `python -m
    torch.distributed.run`, one rank per GPU, rendezvous on 127.0.0.1 -- before torch is imported here.  Rank 0 of the
    child prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks' processes here
    # (the launcher would pin every rank to ONE OpenMP thread: the host side -- k-means++ seeding, Hungarian chain -- gets its share of the cores)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rank_env(args):
    """(rank, local_rank, world) of this process; a launcher's WORLD_SIZE must agree with --gpus."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    return rank, local_rank, world


def launch_check(args):
    import torch
    import torch.distributed as dist

    rank, local_rank, world = rank_env(args)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_counted": int(t.item()), "backend": "gloo"}))
    dist.destroy_process_group()


def newest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.launch_check:
        return launch_check(args)
    cfg = dict(WORKLOADS[args.workload])
    if args.workload == "swfdmc":
        return run_swfdmc(args, cfg)
    W, ell, k, dims = cfg["W"], cfg["ell"], cfg["k"], cfg["dims"]
    D, M = sum(dims), len(dims)

    import torch
    import torch.distributed as dist

    rank, local_rank, world = rank_env(args)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # rehearsal knobs (a one-GPU box cannot host two RCCL ranks): MUSED_DIST_BACKEND=gloo keeps the
    # collectives on CPU tensors, MUSED_FORCE_DEVICE=0 puts every rank on one device
    backend = os.environ.get("MUSED_DIST_BACKEND", "nccl")
    if "MUSED_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["MUSED_FORCE_DEVICE"])
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from mused_amd import distributed as mdist
    from mused_amd import matrix_operations as mo
    from mused_amd.pipeline import StreamPipeline

    K, Wu = args.steps, args.warmup
    # ---- stream layout: lane count is a property of the workload, not of --steps -------------------------------
    B = max(1, min(args.lanes if args.lanes > 0 else cfg["lanes"], K))
    g0 = rank * K                                      # first timed window of this rank (global stream index)
    blocks = [mdist.block_partition(K, B, p) for p in range(B)]
    blks = [b1 - b0 for b0, b1 in blocks]              # timed windows per lane (differ by at most one)
    blk = max(blks)
    T = Wu + blk                                       # lock-steps (a shorter lane repeats its last window: padding)
    padded_slots = sum(blk - b for b in blks)

    def lane_window_index(p, t):
        """global stream index of lock-step t of lane p (t < Wu: its warm-up / halo windows)"""
        s = g0 + blocks[p][0]
        idx = s - Wu + min(t, Wu + blks[p] - 1)
        return idx if idx >= 0 else PRE_STREAM - idx
    host = {}
    for p in range(B):
        for t in range(T):
            gi = lane_window_index(p, t)
            if gi not in host:
                host[gi] = window_rows(cfg, args.kind, gi, args.seed)
    rows_all = torch.empty((B, T, W, D), dtype=torch.float32, device="cuda")   # resident before timing
    for p in range(B):
        for t in range(T):
            rows_all[p, t].copy_(torch.from_numpy(host[lane_window_index(p, t)][0]))
    labels = [[host[lane_window_index(p, t)][1] for t in range(T)] for p in range(B)]

    sketches, grp = [], []   # one sketch object per group of lanes; grp[g] = (first lane, one past the last lane)
    R = 0.0
    if not args.no_swfd:
        from mused_amd.swfd import SeqBasedSWFD

        # R (main.py:61 analogue for the feature sketch) is fixed by window 0 of the stream: rank 0 owns it
        if rank == 0:
            x0 = torch.from_numpy(window_rows(cfg, args.kind, 0, args.seed)[0]).cuda()
            R0 = float((x0.double() ** 2).sum(dim=1).max().item())
            del x0
        else:
            R0 = 0.0
        R = mdist.broadcast_scalar(R0, 0, device=coll_dev) if world > 1 else R0
        G = args.sketch_groups if args.sketch_groups > 0 else (3 if B >= 6 else (2 if B >= 4 else 1))
        G = max(1, min(G, B))
        l0 = 0
        for g in range(G):
            l1 = l0 + B // G + (1 if g < B % G else 0)
            grp.append((l0, l1))
            sketches.append(SeqBasedSWFD(N=W, R=R, d=D, sketch_dim=ell, lanes=l1 - l0))
            l0 = l1
    sketch = sketches[0] if sketches else None
    # different priorities -> different HIP hardware queues (two default-priority streams can land on the
    # same queue and then run strictly in order); the adjacency / eigenstep stream gets the HIGH priority: a chain of
    # ~1600 small dependent launches per window, each of which would otherwise queue behind a full wave of sketch
    # workgroups
    hi_main = os.environ.get('MUSED_BENCH_PRIO', 'main') == 'main'
    st_main = torch.cuda.Stream(priority=-1 if hi_main else 0)
    st_sketch = [torch.cuda.Stream(priority=0 if hi_main else -1) for _ in sketches]
    pipe = StreamPipeline(W, ell, k, args.seed, "sSVDMC", modality_types=[""] * M, feature_sketch=False,
                          async_labels=True, stream=st_main, assume_finite=True,
                          # without the sketch the main path has the GPU to itself: overlap consecutive windows
                          window_slots=int(os.environ.get("MUSED_WINDOW_SLOTS", "4" if args.no_swfd else "1")))
    torch.cuda.synchronize()

    import threading

    sk_events = {}   # (group, t) -> (event, enqueue time)
    sk_out = {}
    refs = [{"ev": None, "t": 0.0} for _ in sketches]
    trace_on = bool(os.environ.get("MUSED_BENCH_TRACE"))

    def drive_sketch(g, lo, hi):
        torch.cuda.set_device(local_rank)  # the current device is per host thread
        ref = refs[g]
        l0, l1 = grp[g]
        with torch.cuda.stream(st_sketch[g]):
            ref["ev"] = torch.cuda.Event(enable_timing=True)
            ref["ev"].record()
            ref["t"] = time.perf_counter()
            for t in range(lo, hi):
                t_enq = time.perf_counter()
                sketches[g].fit_lanes(rows_all[l0:l1, t])
                sk_out[(g, t)] = sketches[g].get_device()
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                sk_events[(g, t)] = (ev, t_enq)
            if hi > lo:
                sk_events[(g, hi - 1)][0].synchronize()
        if trace_on:
            print(f"[trace] sketch group {g} done {time.perf_counter() - ref['t']:.3f}s after its start", file=sys.stderr)

    def trigger_of(p, t):
        return (lane_window_index(p, t) + 1) * W - 1

    def drive_main(lo, hi):
        torch.cuda.set_device(local_rank)
        for t in range(lo, hi):
            for p in range(B):
                if t < Wu + blks[p]:
                    pipe.process_window(split_mods(cfg, rows_all[p, t]), labels[p][t], trigger=trigger_of(p, t))
        pipe.flush()

    def run_range(lo, hi):
        ths = [threading.Thread(target=drive_sketch, args=(g, lo, hi)) for g in range(len(sketches))]
        ths.append(threading.Thread(target=drive_main, args=(lo, hi)))
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    for sk in sketches:
        sk.profile(True)        # (whole-script launch average, for comparison with rocprofv3 --stats of this command)
    run_range(0, Wu)
    other_reads = [sk.profile_read() for sk in sketches]   # warm-up windows
    n_warm_lat = len(pipe.latencies)
    pipe.eng.score_events = []  # HIP events around every similarity-GEMM launch of the timed region
    for sk in sketches:
        sk.profile(True)        # HIP events around every Jacobi sweep graph of the timed region

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_range(Wu, T)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # ---- the halo: every lane first sketches ONE window it does not own (SWFD MAIN(t) continues AUX(t - 1)); in the layout
    # above the warm-up windows play that role.  One sketch-only lock-step of all lanes is timed here, with the same
    # bracket as the timed region, and charged to `value` in proportion: on the 100-window-per-GPU stream of BASELINE
    # config 2 the K timed windows carry K / 100 of it.
    t_halo = 0.0
    if sketches:
        for sk in sketches:
            sk.profile(False)   # (the events of the timed region stay readable)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        ths = [threading.Thread(target=drive_sketch, args=(g, T - 1, T)) for g in range(len(sketches))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_halo = time.perf_counter() - th0
        thmax = torch.tensor([t_halo], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(thmax, op=dist.ReduceOp.MAX)
        t_halo = float(thmax.item())

    # label chain across ranks (outside the timed region: W ints per window), in stream order
    by_trigger = {tr["trigger"]: tr["raw"] for tr in pipe.trace}
    raw_local = np.array([by_trigger[trigger_of(p, t)] for p in range(B) for t in range(Wu, Wu + blks[p])], dtype=np.int64)
    raw_all = mdist.gather_raw_labels(raw_local, [K] * world, device=coll_dev)
    all_labels = mdist.replay_label_chain(raw_all, mo.match_clusters)

    if rank == 0:
        lat = np.array(pipe.latencies[n_warm_lat:])
        if sketches and all(r["ev"] is not None for r in refs):
            # a window is done when its labels AND its sketch are: take the later one.  Lock-step step t of group g
            # completes at t_ref + elapsed(ref event -> its event); latency counts from its enqueue.
            gof = {p: g for g, (a, b) in enumerate(grp) for p in range(a, b)}
            sk_done = {(g, t): refs[g]["t"] + refs[g]["ev"].elapsed_time(sk_events[(g, t)][0]) * 1e-3
                       for g in range(len(sketches)) for t in range(Wu, T)}
            sk_lat = np.array([sk_done[(gof[p], t)] - sk_events[(gof[p], t)][1]
                               for t in range(Wu, T) for p in range(B) if t < Wu + blks[p]])
            lat = np.maximum(lat, sk_lat) if len(lat) == len(sk_lat) else lat
        ev_pairs = pipe.eng.score_events
        pipe.eng.score_events = None
        gemm_live_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else None
        # the live Jacobi timing covers the timed region only: read it before the stand-alone runs below
        osj_reads, trd_reads = [], []
        for sk in sketches:
            osj_reads.append(sk.profile_read())
            trd_reads.append(sk.profile_read_direct())
            sk.profile(False)

        # ---- what ONE stream consumed in order gets: a single-lane sketch beside the main path, window by window ----
        single = None
        if sketches and not args.no_single_lane:
            from mused_amd.swfd import SeqBasedSWFD

            one = SeqBasedSWFD(N=W, R=R, d=D, sketch_dim=ell, lanes=1)   # (its rotations use the persistent osjq_kernel)
            nwin = min(T, 4)
            lats = []
            for t in range(T - nwin, T):   # the first is a warm-up (graph upload) and the sketch's halo window
                ts = time.perf_counter()

                def _sk():
                    torch.cuda.set_device(local_rank)
                    with torch.cuda.stream(st_sketch[0]):
                        one.fit(rows_all[0, t])
                        one.get_device()
                        st_sketch[0].synchronize()
                th = threading.Thread(target=_sk)
                th.start()
                pipe.process_window(split_mods(cfg, rows_all[0, t]), labels[0][t], trigger=-(t + 1))
                pipe.flush()
                th.join()
                lats.append(time.perf_counter() - ts)
            one.close()
            p50 = float(np.median(lats[1:])) if len(lats) > 1 else float(lats[0])
            single = {"rows_per_s": W / p50, "p50_window_ms": 1e3 * p50, "windows_timed": max(1, len(lats) - 1),
                      "note": "one window at a time, strictly in order: a 1-lane sketch on its stream beside adjacency -> "
                              "eigenstep -> labels of the same window; nothing batched across windows"}

        if sketch is not None:
            sketch.profile(True)
        stages = stage_profile(cfg, rows_all[0, -1], pipe, sketch, rows_all[grp[0][0]:grp[0][1]] if grp else rows_all)
        if sketch is not None:
            other_reads.append(sketch.profile_read())   # stand-alone stage run of group 0
            sketch.profile(False)
        stages["swfd_groups"] = [b - a for a, b in grp]
        stages["scores_gemm_ms_live_timed_region"] = gemm_live_ms
        L_sk = stages.get("swfd_levels") or 0

        # ---- rooflines (measured live with HIP events on the launch streams over the timed region) ----
        # (1) dominant kernel by time: osjw_kernel, one block-pair round of the one-sided Jacobi of the FD rotation.
        #     It contains no MFMA: a VALU-issue-bound chain of dependent pair-steps between a load and a store phase
        #     (DESIGN.md section 5).  Priced on its executed fp64 flops against the fp64 rate of the chip (vector FMA and
        #     MFMA peaks coincide at 78.6 TF/s on gfx950); `roofline_hbm` prices the same launches on the bytes they move.
        roof = roof_hbm = None
        n2 = 2 * ell
        # one sweep = n (n - 1) / 2 column pairs x (2 n flop dot product + 4 n flop rotation), spread over nb - 1 launches
        # (orders <= 256; nb = n / 32 column blocks) or nb launches (orders 320-512: one more for the pairs inside the blocks)
        nb = max(2, -(-n2 // 64) * 2)
        flops_per_matrix_launch = (n2 * (n2 - 1) // 2) * (6.0 * n2) / (nb - 1 if n2 <= 256 else nb)
        direct = bool(sketches) and all(r[0] for r in trd_reads)
        if direct:
            # dominant kernel by time: trd_a_kernel (csrc/trd.hip), the Householder tridiagonalisation of the direct eigensolver
            # of the FD rotation -- one workgroup (one CU) per Gram matrix of order 256, the matrix in registers, fp64 vector
            # ALU; what bounds it is the chain of dependent steps (4 workgroup barriers per column, 254 columns), not bytes.
            # The solver is a chain of five kernels (trd_a tridiagonalisation, trd_b 128 eigenvalues by multisection on sign
            # counts, trd_c their vectors by twisted factorisation, trd_t + trd_d back-transformation on the matrix cores):
            # `roofline` prices trd_a alone on its own HIP-event bracket, `roofline.solve` the whole chain.
            n = n2
            m_top = ell
            flop_solve = {
                "trd_a_tridiagonalisation_4n3_3": 4.0 * n ** 3 / 3.0,
                "trd_t_trd_d_back_transformation_2n2m_mfma": 2.0 * n * n * m_top,
                "trd_b_sign_counts_20_per_thread_x_512_threads_x_4flop": 20 * 512 * (n - 1) * 4.0,
                "trd_c_twisted_factorisation": m_top * 4.0 * (n - 1) * 6.0,
            }
            fl = sum(flop_solve.values())
            fl_a = flop_solve["trd_a_tridiagonalisation_4n3_3"]
            t_ms = sum(r[1] for r in trd_reads)
            n_launch = sum(r[2] for r in trd_reads)
            solved = sum(r[3] for r in trd_reads)
            ta_ms = sum(r[4] for r in trd_reads)
            ns = len(sketches)
            if n_launch and t_ms > 0 and ta_ms > 0:
                launch_us = 1e3 * t_ms / n_launch
                a_us = 1e3 * ta_ms / n_launch
                per_launch = solved / n_launch
                tfl = per_launch * fl / (launch_us * 1e-6) / 1e12
                tfl_a = per_launch * fl_a / (a_us * 1e-6) / 1e12
                tr = tr_a = None
                try:
                    pmf = newest_profile("r*_pmc_trd.json")
                    if pmf:
                        pm = json.load(open(pmf))
                        tr = pm["traffic_bytes_per_matrix"] * per_launch
                        tr_a = pm["per_kernel"]["trd_a_kernel"]["traffic_bytes_per_matrix"] * per_launch
                except Exception:
                    tr = tr_a = None
                cu_peak = 1e3 * FP64_PEAK_TFLOPS / 256
                roof = {
                    "kernel": f"trd_a_kernel (Householder tridiagonalisation of the direct symmetric eigensolver of the FD rotation: one "
                              f"workgroup = one CU per Gram matrix of order {n}, matrix in registers, fp64 vector ALU), "
                              f"{np.mean([sk.lanes * 2 * sk.L for sk in sketches]):.0f} matrices per launch of which {per_launch:.1f} "
                              f"are solved (duplicates / frozen sketches skipped), {ns} independent launch streams",
                    "bound": "mfma",
                    "bound_detail": "fp64 vector ALU (no MFMA in this kernel); priced against the fp64 rate of the chip, 78.6 TFLOP/s, "
                                    "which on gfx950 is the fp64 MFMA peak as well",
                    "achieved": tfl_a,
                    "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": tfl_a / FP64_PEAK_TFLOPS,
                    "traffic": tr_a,
                    "launch_us": a_us,
                    "launches_timed": n_launch,
                    "matrices_solved_per_launch_avg": per_launch,
                    "algorithmic_flops_per_matrix": fl_a,
                    "executed_flops_per_launch": per_launch * fl_a,
                    "per_cu": {"achieved_gflops": fl_a / (a_us * 1e-6) / 1e9, "peak_gflops": cu_peak,
                               "frac": fl_a / (a_us * 1e-6) / 1e9 / cu_peak},
                    "concurrent_launch_streams": ns,
                    "solve": {
                        "kernels": "trd_a_kernel -> trd_b_kernel -> trd_c_kernel -> trd_t_kernel -> trd_d_kernel (one HIP-event bracket)",
                        "launch_us": launch_us, "flop_per_matrix": flop_solve, "achieved": tfl, "frac": tfl / FP64_PEAK_TFLOPS,
                        "per_cu_frac": fl / (launch_us * 1e-6) / 1e9 / cu_peak, "traffic": tr,
                    },
                    "note": "`bound`: the contract's enum is hbm | mfma; this kernel is compute, not bytes -- fp64 VALU work in a chain of "
                            "dependent, barrier-separated steps (one CU per matrix), priced on 4 n^3 / 3 flops per matrix against "
                            "the fp64 vector rate of the chip (78.6 TFLOP/s, the same number as the fp64 MFMA peak).  "
                            "A launch occupies `matrices_solved_per_launch_avg` of the 256 CUs: `per_cu` is what one busy CU "
                            "reaches.  `launch_us` is the HIP-event time from the start of the solver chain to the end of "
                            "trd_a_kernel on its launch stream, with the other streams of the pipeline running beside it (the "
                            "rocprofv3 average of trd_a_kernel under profiles/ is the same quantity).  Algorithmic bytes per "
                            "matrix: 256 KB read (lower triangle of G) + 512 KB of Householder vectors written.",
                }
                if n > 256:
                    # orders 320 .. 1024 run the BLOCKED solver (csrc/trdx.hip): its tridiagonalisation streams the lower triangle of
                    # the matrix once per column -- bytes, not flops, are what it is priced on.  It reduces the first ldn - 256
                    # columns; the trailing 256 x 256 goes to the register-resident trd_a_kernel.
                    ldn = -(-n // 64) * 64 if n <= 512 else -(-n // 128) * 128
                    jend = ldn - 256
                    # symv stream of columns 0 .. jend - 1 + the matrix read once + Householder vectors of those columns + the tail block
                    bytes_a = (ldn ** 3 - 256 ** 3) // 6 * 8 + ldn * ldn * 8 + jend * ldn * 8 + 256 * 256 * 8
                    gbs_a = per_launch * bytes_a / (a_us * 1e-6) / 1e9
                    trx = None
                    try:
                        pmf = newest_profile("r*_pmc_trdx.json")
                        if pmf:
                            pm = json.load(open(pmf))
                            trx = pm["per_kernel"]["trdx_a_kernel"]["traffic_bytes_per_matrix"] * per_launch
                    except Exception:
                        trx = None
                    roof = {
                        "kernel": f"trdx_a_kernel (blocked Householder tridiagonalisation of the direct eigensolver of the FD rotation, "
                                  f"order {ldn}: one workgroup per Gram matrix, dlatrd panels of 16 columns over the first {jend} columns -- the "
                                  f"symv streams the lower triangle of the panel-start matrix from L2 / Infinity Cache / HBM once per "
                                  f"column -- the trailing 256 x 256 handed to the register-resident trd_a_kernel), {per_launch:.1f} "
                                  f"matrices solved per launch, {ns} independent launch streams",
                        "bound": "hbm", "achieved": gbs_a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_a / HBM_PEAK_GBS,
                        "traffic": trx, "launch_us": a_us, "launches_timed": n_launch,
                        "matrices_solved_per_launch_avg": per_launch, "algorithmic_bytes_per_matrix": bytes_a,
                        "per_cu": {"achieved_gbs": bytes_a / (a_us * 1e-6) / 1e9,
                                   "note": "one workgroup = one CU per matrix: what a single CU pulls (64 B/clk of L1 = 150 GB/s is its ceiling)"},
                        "concurrent_launch_streams": ns,
                        "solve": {"kernels": "trdx_a -> trd_a (trailing 256 x 256) -> trdx_tail_merge -> trd_b -> trd_c -> trdx_cert -> "
                                             "trdx_tfac -> trdx_back -> trdx_store (one HIP-event bracket)", "launch_us": launch_us,
                                  "flop_per_matrix": flop_solve, "achieved_tflops": tfl},
                        "note": "algorithmic bytes = (n^3 - 256^3) / 6 doubles of symv reads (each of the first n - 256 columns multiplies the "
                                "trailing lower triangle) + n^2 doubles read (working copy) + the Householder vectors and the tail block "
                                "written; `launch_us` is the HIP-event time from "
                                "the start of the solver chain to the end of trdx_a_kernel on its launch stream with the other streams of "
                                "the pipeline running beside it.  The kernel is latency-bound on the one tile (16 KB) a wave keeps in "
                                "flight (DESIGN section 5b), far from the HBM roofline.",
                    }
                roof_hbm = {
                    "kernel": "the whole solver chain priced on bytes: G read once (512 KB), 128 columns written (256 KB) per matrix",
                    "bound": "hbm", "achieved": per_launch * 786432.0 / (launch_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": per_launch * 786432.0 / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": tr, "launch_us": launch_us,
                }
        if sketches and not direct:
            osj_ms = osj_launches = 0
            osj_total_bytes = 0.0
            for ms_g, n_g, b_g in osj_reads:
                osj_ms += ms_g
                osj_launches += n_g
                osj_total_bytes += n_g * b_g
            if osj_launches:
                osj_us = 1e3 * osj_ms / osj_launches
                osj_bytes = osj_total_bytes / osj_launches
                gbs = osj_bytes / (osj_us * 1e-6) / 1e9
                mats_per_launch = np.mean([sk.lanes * 2 * sk.L for sk in sketches])
                per_matrix = 16.0 * n2 ** 2                 # every matrix read + written once per launch
                active_mats = osj_bytes / per_matrix        # < matrices per launch: adaptive sweep count, duplicates skipped
                tfl = active_mats * flops_per_matrix_launch / (osj_us * 1e-6) / 1e12
                tr = None
                try:
                    pmf = newest_profile("r*_pmc_osj.json")
                    if args.workload in ("c2", "c4") and pmf:   # PMC passes: every matrix active in every launch -> bytes per matrix
                        pm = json.load(open(pmf))
                        tr = pm["traffic_bytes_per_launch"] / pm["matrices_per_launch"] * active_mats
                except Exception:
                    tr = None
                ns = len(sketches)
                roof = {
                    "kernel": f"osjw_kernel<{max(1, -(-n2 // 64))}> (block-pair round of the one-sided Jacobi of the FD rotation; fp64 "
                              f"vector FMA, no MFMA), {mats_per_launch:.0f} Gram matrices of order {n2} per launch, "
                              f"{ns} independent launch streams",
                    "bound": "mfma",
                    "achieved": tfl,
                    "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": tfl / FP64_PEAK_TFLOPS,
                    "traffic": tr,
                    "launch_us": osj_us,
                    "launches_timed": osj_launches,
                    "executed_flops_per_launch": active_mats * flops_per_matrix_launch,
                    "matrices_active_per_launch_avg": active_mats,
                    "concurrent_launch_streams": ns,
                    "achieved_all_streams": tfl * ns,
                    "frac_all_streams": tfl * ns / FP64_PEAK_TFLOPS,
                    "note": "`bound` names the fp64 arithmetic rate (the contract's enum has no VALU entry): the kernel is "
                            "VALU-issue bound -- 96 of ~210 vector instructions per pair-step are the FMAs counted here "
                            "(dot products 2n + rotation 4n flop per column pair); per-launch figures, launches of the "
                            "sketch groups run side by side (`achieved_all_streams`)",
                }
                roof_hbm = {
                    "kernel": "same launches priced on bytes: every active matrix read once and written once per launch",
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "traffic": tr, "launch_us": osj_us, "algorithmic_bytes_per_launch": osj_bytes,
                    "achieved_all_streams": gbs * ns, "frac_all_streams": gbs * ns / HBM_PEAK_GBS,
                }
        # (2) the contraction kernel: similarity GEMM X X^T on fp64 MFMA
        # SURVEY 8(d) counts 2 W d flop per row x W rows; the kernel computes the tiles on or above the diagonal only
        # (the distance / cosine epilogue is symmetric) and writes each of them twice: `achieved` is priced on the MFMA
        # work actually executed, `algorithmic_flops_per_launch` keeps the SURVEY figure
        d0 = dims[0]
        flops = 2.0 * W * W * d0
        nt = -(-W // 128)
        flops_exec = 2.0 * d0 * 128.0 * 128.0 * (nt * (nt + 1) // 2)
        gemm_ms = gemm_live_ms if gemm_live_ms else stages["scores_gemm_ms"]
        gemm_s = gemm_ms * 1e-3
        traffic = None
        try:  # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/), config 2 only
            pmf = newest_profile("r*_pmc_scores.json")
            if args.workload == "c2" and pmf:
                traffic = json.load(open(pmf))["traffic_bytes_per_launch_lower"]
        except Exception:
            traffic = None
        fused_knn = pipe.eng.knn_mode != "classic"
        roof_gemm = {
            "kernel": ("knn_band_kernel<float> x 4 phases + cand_select_kernel (pairwise squared distances on "
                       "v_mfma_f64_16x16x4_f64, symmetric tile grid walked by cyclic tile distance, top-k candidates "
                       "filtered in the epilogue: no score matrix; the time covers the whole call incl. the selections)")
            if fused_knn else
                      ("gemm_f64_kernel<float,float,NT> + EpiSqL2 (pairwise squared distances, v_mfma_f64_16x16x4_f64, "
                       "upper-triangular tiles + mirrored stores)"),
            "bound": "mfma",
            "achieved": flops_exec / gemm_s / 1e12,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": flops_exec / gemm_s / 1e12 / FP64_PEAK_TFLOPS,
            "traffic": traffic,
            "launch_ms": gemm_ms,
            "launch_ms_standalone": stages["knn_fused_ms"] if fused_knn else stages["scores_gemm_ms"],
            "executed_flops_per_launch": flops_exec,
            "algorithmic_flops_per_launch": flops,
            "algorithmic_tflops_equivalent": flops / gemm_s / 1e12,
        }
        if roof is not None and not direct:
            # every Jacobi launch of this script (warm-up windows, timed region, stand-alone stage run):
            # the population a `rocprofv3 --kernel-trace --stats -- python3 bench.py` average is taken over
            ms_all = osj_ms + sum(r[0] for r in other_reads)
            n_all = osj_launches + sum(r[1] for r in other_reads)
            roof["launch_us_whole_script"] = 1e3 * ms_all / n_all if n_all else None
            roof["launches_whole_script"] = n_all
        if roof is None:
            roof = roof_gemm
        value_excl_halo = world * K * W / elapsed
        halo_share = t_halo * K / 100.0          # K of the 100 windows a GPU owns in BASELINE config 2
        value = world * K * W / (elapsed + halo_share)
        # (3) the whole path in SURVEY 8(d) units: algorithmic flop per row (SWFD 12 l d per FD instance, similarity
        #     2 W d per modality, eigenstep 13 * 2 W (l + 10), dense variant) x rows/s against the fp64 peak
        per_row = 0.0
        if sketches:
            per_row += 12.0 * ell * D * (2 * L_sk)
        per_row += sum(2.0 * W * dm for dm in dims)
        nnz = W * (k - 1) * M
        eig_bytes = 13.0 * nnz * (ell + 10) * 8.0          # SURVEY 8(d), neighbour-list SpMM variant (the one that runs)
        roof_8d = {
            "definition": "SURVEY 8(d) algorithmic work per row x measured rows/s of one GPU: SWFD + similarity in flop against "
                          "the fp64 peak; the eigenstep (13 SpMM by neighbour lists) in bytes against HBM, on its own stage time",
            "mflop_per_row": per_row / 1e6,
            "swfd_fd_instances": 2 * L_sk if sketches else 0,
            "achieved": per_row * (value / world) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": per_row * (value / world) / 1e12 / FP64_PEAK_TFLOPS,
            "eigenstep": {"bound": "hbm", "algorithmic_bytes_per_window": eig_bytes, "stage_ms_alone": stages["rsvd_ms"],
                          "achieved": eig_bytes / (stages["rsvd_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": eig_bytes / (stages["rsvd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        sha16 = hashlib.sha256(all_labels.astype(np.int64).tobytes()).hexdigest()[:16]
        golden_ok = None
        try:   # event labels of the REFERENCE's own window loop over the same windows (tests/golden/make_golden.py)
            gp = os.path.join(ROOT, "tests", "golden", f"bench_{args.workload}_{args.kind}_s{args.seed}.npz")
            if os.path.exists(gp):
                gg = np.load(gp, allow_pickle=False)
                if world * K <= int(gg["meta"][0]):
                    golden_ok = bool(str(gg["cumulative_sha16"][world * K - 1]) == sha16)
        except Exception:
            golden_ok = None
        res = {
            "metric": ("stream rows/sec, d=1024 l=128 window=10k synthetic (SWFD + kNN similarity + eigenstep + labels)"
                       if args.workload in ("c2", "c4") else f"stream rows/sec, {cfg['name']} (SWFD + kNN similarity + eigenstep + labels)"),
            "value": value,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wu,
            "ms_per_step": 1e3 * (elapsed + halo_share) / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": cfg["name"],
                "stream": args.kind,
                "W": W, "d": D, "l": ell, "k": k, "modalities": M,
                "swfd_levels": L_sk,
                "parallelism": f"windows sharded in contiguous blocks over {world} GPU(s) x {B} lock-step lane(s) per GPU",
                "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                "collective_backend": (dist.get_backend() if world > 1 else None),
                "lanes_per_gpu": B,
                "padded_window_slots": padded_slots,
                "labels_sha16": sha16,
                "labels_match_reference_golden": golden_ok,
            },
            "p50_window_latency_ms": float(np.median(lat) * 1e3) if len(lat) else None,
            "single_lane": single,
            # every lane re-sketches ONE window it does not own (its halo: SWFD MAIN(t) continues AUX(t-1)); the warm-up
            # windows play that role here and are not timed.  On the 100-window-per-GPU stream of BASELINE config 2
            # (1M rows) B lanes cost B extra sketch windows:
            "value_excl_halo": value_excl_halo,
            "halo": {"lock_step_s": t_halo, "share_charged_s": halo_share, "timed_s": elapsed,
                     "rule": "value = rows / (timed + halo lock-step x K / 100): every lane sketches one window it does not own "
                             "before its block; on the 100-window-per-GPU stream of BASELINE config 2 the K timed windows "
                             "carry K / 100 of that lock-step (measured here: all lanes, sketch only, same bracket)"},
            "stages_ms": stages,
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "roofline_mfma": roof_gemm,
            "roofline_8d": roof_8d,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.kind, args.seed, with_swfd=not args.no_swfd)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    pipe.close()
    for sk in sketches:
        sk.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
