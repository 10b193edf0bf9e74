#!/usr/bin/env python3
"""Headline benchmark: stream rows/s (+ p50 per-window latency) of the MI355X hot path on the
BASELINE.json config-2 workload: synthetic stream, d = 1024, l = 128, window W = N = 10,000, k = 50.

One STEP = one window of W rows, already resident in HBM, through the whole path:
    SeqBasedSWFD over the W feature rows (append + get)         [a5-a7]
    Euclidean kNN adjacency of the window (fp64 MFMA scores + exact selection)   [a1]
    fusion / R                                                  [a3, a4]
    randomized-SVD eigenstep on the fused adjacency             [a8]
    k-means + Hungarian matching on the host (sklearn / SciPy)  [a10]  -> event labels
`value` = rows of all ranks / wall time of the K timed steps (max over ranks), inputs resident.

    python bench.py                       # 1 GPU, K = 5, W = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: every rank owns a contiguous block of windows of the stream (mused_amd/distributed.py):
weak scaling, no data-path collective; one all-gather of raw labels at the end.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration `metric` is quoted on
    "c2": dict(W=10000, d=1024, ell=128, k=50, name="synthetic d=1024 l=128 window=10000 k=50 (BASELINE config 2)"),
    # small plumbing case (configs[0] shapes) for quick checks
    "c1": dict(W=500, d=64, ell=16, k=50, name="synthetic d=64 l=16 window=500 k=50 (BASELINE config 1)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--kind", default="blob", choices=["blob", "gauss", "fd"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-swfd", action="store_true", help="diagnostic: skip the feature-row SWFD stage")
    ap.add_argument("--lanes", type=int, default=9,
                    help="contiguous blocks of the rank's windows whose sketches advance in lockstep inside the same "
                         "launches (1 = strictly one window at a time)")
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


def stage_profile(cfg, X, pipe, sketch, rows_all):
    """Per-stage device times (ms, HIP events on the launch stream) for one resident window, plus
    the live roofline measurement of the dominant kernels."""
    import ctypes as C

    import torch

    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    W, d, ell, k = cfg["W"], cfg["d"], cfg["ell"], cfg["k"]
    eng = pipe.eng
    st = torch.cuda.current_stream()
    out = {}
    # similarity GEMM alone: row norms + fp64 MFMA X X^T with the distance epilogue
    dt = _lib.F32 if X.dtype == torch.float32 else _lib.F64
    f_scores = lambda: _lib.call("mused_pairwise_scores", ptr(X), dt, W, d, X.stride(0), 0, ptr(eng.norms),
                                 ptr(eng.scores), stream_ptr())
    f_scores()
    out["scores_gemm_ms"] = hip_event_ms(f_scores, st, 3)
    w = (W + 63) // 64
    mask = torch.empty((W, w), dtype=torch.int64, device="cuda")
    f_sel = lambda: _lib.call("mused_select_k_smallest", ptr(eng.scores), W, W, k, None, ptr(mask), w, stream_ptr())
    f_sel()
    out["select_ms"] = hip_event_ms(f_sel, st, 3)
    adj = eng.knn_adjacency(X, k)
    f_rsvd = lambda: eng.svd_reduce(adj, ell, pipe.seed, nnz_cap=W * k)
    f_rsvd()
    out["rsvd_ms"] = hip_event_ms(f_rsvd, st, 2)
    if sketch is not None:
        f_app = lambda: sketch.fit_lanes(rows_all[:, -1])
        out["swfd_append_ms_all_lanes"] = hip_event_ms(f_app, st, 1)
        f_get = lambda: sketch.get_device()
        out["swfd_query_ms_all_lanes"] = hip_event_ms(f_get, st, 1)
        out["swfd_levels"] = sketch.L
        out["swfd_lanes"] = sketch.lanes
    if sketch is not None and os.environ.get("MUSED_BENCH_LATENCY_PROBE"):
        # latency-oriented setting for comparison (not the throughput configuration that is timed; off by default so
        # that a rocprofv3 --stats run of this script averages the same launches as the live timing): ONE window at a
        # time through a single-lane sketch -- what a window costs when nothing is batched across windows
        from mused_amd.swfd import SeqBasedSWFD

        one = SeqBasedSWFD(N=W, R=sketch.R, d=d, sketch_dim=ell, lanes=1)
        one.fit(X)
        f_one = lambda: (one.fit(X), one.get_device())
        out["swfd_window_ms_single_lane"] = hip_event_ms(f_one, st, 1)
        one.close()
    return out


def cpu_baseline(cfg, kind, seed, with_swfd=True):
    """The CPU oracle (oracle/*.py, a port of the reference path pinned to its golden vectors) timed
    on this box's host cores over a bounded sample of the same workload.  BLAS / OpenMP pools are capped at
    16 threads (a one-GPU share of the host): left at one thread per visible core (256 here) the same sample
    runs ~10x slower, which would flatter the GPU."""
    from mused_amd import synth
    from oracle import mo_oracle as omo
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    threads = max(1, min(16, os.cpu_count() or 1))
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter, threads = None, os.cpu_count() or 1
    W, d, ell, k = cfg["W"], cfg["d"], cfg["ell"], cfg["k"]
    X, labels = synth.stream_window(kind, 0, W, d, seed)
    t0 = time.perf_counter()
    A = omo.create_adjacency_matrix(X.astype(np.float64), "", k)
    F = omo.fuse_matrices([A])
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
        sk = OraSWFD(N=W, R=R, d=d, sketch_dim=ell)
        swfd_rows = min(W, 4 * ell)  # 4 rotations of every level; steady-state cost per row is constant
        ts = time.perf_counter()
        sk.fit(X64[:swfd_rows])
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
                  f"k-means ({t3 - t2:.2f}s); SWFD oracle timed on {swfd_rows} rows "
                  f"({t_swfd_per_row * 1e3:.2f} ms/row) and extrapolated to the window; BLAS/OpenMP pools capped at "
                  f"{threads} threads",
        "window_seconds": window_s,
    }


def main():
    args = parse()
    cfg = dict(WORKLOADS[args.workload])
    W, d, ell, k = cfg["W"], cfg["d"], cfg["ell"], cfg["k"]

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from mused_amd import synth
    from mused_amd.pipeline import StreamPipeline

    K, Wu = args.steps, args.warmup
    # Window-level concurrency on one GPU: the rank's K timed windows are split into B contiguous
    # blocks ("lanes").  The B sketch sets advance in LOCKSTEP inside the same kernel launches
    # (mused_swfd_*_lanes) on one HIP stream / host thread; adjacency + eigenstep + labels of the same
    # windows run on a second stream / host thread.  Every block is preceded in the stream by its Wu
    # warm-up windows, which double as the SWFD halo (mused_amd/distributed.py).
    # lanes: at most --lanes; the count that minimises (lock-step groups) x (time of a group of B lanes).  A group
    # is ~78 rotations x ~72 Jacobi launches, and a launch runs 8 L B workgroups (L levels, MAIN + AUX, 4 block
    # pairs) over 512 resident slots (256 CUs x 2): its time is a step function of B -- B = 9 fills the
    # second round that B = 5..8 leave partly empty (measured: 12 + 43 x rounds microseconds per launch).
    w0 = synth.stream_window(args.kind, 0, W, d, args.seed)[0]  # window 0 of the stream fixes R
    L_est = max(1, int(np.ceil(np.log2(max(float((w0.astype(np.float64) ** 2).sum(1).max()), 1.0))))) + 1
    n_rot = -(-W // ell)
    group_ms = lambda b: 60.0 + 10.0 * b + n_rot * 72 * (12.0 + 43.0 * (-(-8 * L_est * b // 512))) * 1e-3
    cand = range(1, max(1, min(args.lanes, K)) + 1)
    B = min(cand, key=lambda b: ((-(-K // b)) * group_ms(b), -b))
    blks = [K // B + (1 if p < K % B else 0) for p in range(B)]   # timed windows per lane
    blk = max(blks)
    T = Wu + blk                      # lock-step groups (a lane with fewer windows repeats its last one: padding)
    per_rank = K + B * Wu
    first = rank * per_rank           # global window index of lane 0, window 0
    bases, b0 = [], first
    for p in range(B):
        bases.append(b0)
        b0 += Wu + blks[p]
    host = [[synth.stream_window(args.kind, bases[p] + t, W, d, args.seed) for t in range(Wu + blks[p])] for p in range(B)]
    rows_all = torch.empty((B, T, W, d), dtype=torch.float32, device="cuda")   # resident before timing
    for p in range(B):
        for t in range(T):
            rows_all[p, t].copy_(torch.from_numpy(host[p][min(t, Wu + blks[p] - 1)][0]))
    labels = [[l for _, l in hp] for hp in host]

    sketch = None
    if not args.no_swfd:
        from mused_amd.swfd import SeqBasedSWFD

        # R (main.py:61 analogue for the feature sketch) is fixed by window 0 of the stream: rank 0 owns it
        R0 = float((rows_all[0, 0].double() ** 2).sum(dim=1).max().item()) if rank == 0 else 0.0
        R = mdist.broadcast_scalar(R0, 0, device=coll_dev) if world > 1 else R0
        sketch = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
    # different priorities -> different HIP hardware queues (two default-priority streams can land on the
    # same queue and then run strictly in order)
    # the adjacency / eigenstep stream gets the HIGH priority: a chain of ~3600 small dependent launches per window,
    # each of which would otherwise queue behind a full wave of sketch workgroups
    hi_main = os.environ.get('MUSED_BENCH_PRIO', 'main') == 'main'
    st_sketch, st_main = torch.cuda.Stream(priority=0 if hi_main else -1), torch.cuda.Stream(priority=-1 if hi_main else 0)
    pipe = StreamPipeline(W, ell, k, args.seed, "sSVDMC", feature_sketch=False, async_labels=True, stream=st_main)
    torch.cuda.synchronize()

    import threading

    sk_events = {}
    sk_out = {}

    ref = {"ev": None, "t": 0.0}

    def drive_sketch(lo, hi):
        if sketch is None:
            return
        torch.cuda.set_device(local_rank)  # the current device is per host thread
        with torch.cuda.stream(st_sketch):
            ref["ev"] = torch.cuda.Event(enable_timing=True)
            ref["ev"].record()
            ref["t"] = time.perf_counter()
            for t in range(lo, hi):
                t_enq = time.perf_counter()
                sketch.fit_lanes(rows_all[:, t])
                sk_out[t] = sketch.get_device()
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                sk_events[t] = (ev, t_enq)
            if hi > lo:
                sk_events[hi - 1][0].synchronize()
        if os.environ.get("MUSED_BENCH_TRACE"):
            print(f"[trace] sketch thread done {time.perf_counter() - ref['t']:.3f}s after its start", file=sys.stderr)

    def drive_main(lo, hi):
        # window order of the label chain: lane-major within the rank is restored after the run
        torch.cuda.set_device(local_rank)
        tm0 = time.perf_counter()
        for t in range(lo, hi):
            for p in range(B):
                if t < Wu + blks[p]:
                    pipe.process_window([rows_all[p, t]], labels[p][t], trigger=(bases[p] + t + 1) * W - 1)
                    if os.environ.get("MUSED_BENCH_TRACE"):
                        print(f"[trace] main enqueued window ({t},{p}) at {time.perf_counter() - tm0:.3f}s", file=sys.stderr)
        pipe.flush()
        if os.environ.get("MUSED_BENCH_TRACE"):
            print(f"[trace] main thread done {time.perf_counter() - tm0:.3f}s; label latencies {[round(x, 3) for x in pipe.latencies[-B:]]}"
                  f" kmeans ms {[round(x) for x in pipe.host_ms['kmeans'][-B:]]} match ms {[round(x) for x in pipe.host_ms['match'][-B:]]}", file=sys.stderr)

    def run_range(lo, hi):
        ths = [threading.Thread(target=drive_sketch, args=(lo, hi)), threading.Thread(target=drive_main, args=(lo, hi))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    run_range(0, Wu)
    n_warm_lat = len(pipe.latencies)
    pipe.eng.score_events = []  # HIP events around every similarity-GEMM launch of the timed region
    if sketch is not None:
        sketch.profile(True)    # HIP events around every Jacobi sweep graph of the timed region

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
    S = B

    # label chain across ranks (outside the timed region: W ints per window)
    # traces were appended in (t, lane) order; the stream order is lane-major
    by_trigger = {tr["trigger"]: tr["raw"] for tr in pipe.trace}
    raw_local = np.array([by_trigger[(bases[p] + t + 1) * W - 1] for p in range(B) for t in range(Wu, Wu + blks[p])],
                         dtype=np.int64)
    counts = [K] * world
    raw_all = mdist.gather_raw_labels(raw_local, counts, device=coll_dev)
    all_labels = mdist.replay_label_chain(raw_all, mo.match_clusters)

    if rank == 0:
        lat = np.array(pipe.latencies[n_warm_lat:])
        if sketch is not None and ref["ev"] is not None:
            # a window is done when its labels AND its sketch are: take the later one.  Sketch group t
            # completes at t_ref + elapsed(ref event -> its event); latency counts from its enqueue.
            sk_done = {t: ref["t"] + ref["ev"].elapsed_time(sk_events[t][0]) * 1e-3 for t in range(Wu, T)}
            sk_lat = np.array([sk_done[t] - sk_events[t][1] for t in range(Wu, T) for p in range(B) if t < Wu + blks[p]])
            lat = np.maximum(lat, sk_lat) if len(lat) == len(sk_lat) else lat
        ev_pairs = pipe.eng.score_events
        pipe.eng.score_events = None
        gemm_live_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else None
        stages = stage_profile(cfg, rows_all[0, -1], pipe, sketch, rows_all)
        stages["scores_gemm_ms_live_timed_region"] = gemm_live_ms
        # ---- rooflines (both measured live with HIP events on the launch streams over the timed region) ----
        # (1) dominant kernel by time: osjw_kernel, one block-pair round of the one-sided Jacobi of the FD
        #     rotation.  Per launch it streams every Gram matrix that is still iterating from memory and back
        #     (16 n^2 B per matrix): load/store phases around a VALU-issue-bound chain of
        #     32 (round 0: 63) dependent pair-steps.
        roof = None
        if sketch is not None:
            osj_ms, osj_launches, osj_bytes = sketch.profile_read()
            sketch.profile(False)
            if osj_launches:
                osj_us = 1e3 * osj_ms / osj_launches
                gbs = osj_bytes / (osj_us * 1e-6) / 1e9
                tr = None
                full_bytes = 16.0 * sketch.lanes * 2 * sketch.L * (2 * ell) ** 2   # every matrix read + written once
                active_frac = osj_bytes / full_bytes    # < 1: adaptive sweep count, late launches find fewer matrices
                try:
                    if args.workload == "c2":
                        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_osj.json")))
                        if pm.get("lanes") == B:   # PMC passes ran with every matrix active in every launch
                            tr = pm["traffic_bytes_per_launch"] * active_frac
                except Exception:
                    tr = None
                roof = {
                    "kernel": f"osjw_kernel<{2 * ell // 64}> (block-pair round of the one-sided Jacobi of the FD rotation, "
                              f"{sketch.lanes * 2 * sketch.L} Gram matrices of order {2 * ell} per launch)",
                    "bound": "hbm",
                    "achieved": gbs,
                    "peak": 8000.0,
                    "unit": "GB/s",
                    "frac": gbs / 8000.0,
                    "traffic": tr,
                    "launch_us": osj_us,
                    "launches_timed": osj_launches,
                    "algorithmic_bytes_per_launch": osj_bytes,
                    "matrices_active_per_launch_avg": active_frac * sketch.lanes * 2 * sketch.L,
                }
        # (2) the contraction kernel: similarity GEMM X X^T on fp64 MFMA
        flops = 2.0 * W * W * d  # similarity = 2 W d flop per row x W rows per launch
        gemm_ms = gemm_live_ms if gemm_live_ms else stages["scores_gemm_ms"]
        gemm_s = gemm_ms * 1e-3
        traffic = None
        try:  # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/), config 2 only
            if args.workload == "c2":
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_scores.json")))
                traffic = pm["traffic_bytes_per_launch_lower"]
        except Exception:
            traffic = None
        roof_gemm = {
            "kernel": "gemm_f64_kernel<float,float,NT> + EpiSqL2 (pairwise squared distances, v_mfma_f64_16x16x4_f64)",
            "bound": "mfma",
            "achieved": flops / gemm_s / 1e12,
            "peak": 78.6,
            "unit": "TFLOP/s",
            "frac": flops / gemm_s / 1e12 / 78.6,
            "traffic": traffic,
            "launch_ms": gemm_ms,
            "launch_ms_standalone": stages["scores_gemm_ms"],
            "algorithmic_flops_per_launch": flops,
        }
        if roof is None:
            roof = roof_gemm
        res = {
            "metric": "stream rows/sec, d=1024 l=128 window=10k synthetic (SWFD + kNN similarity + eigenstep + labels)",
            "value": world * K * W / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wu,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": cfg["name"],
                "stream": args.kind,
                "W": W, "d": d, "l": ell, "k": k, "modalities": 1,
                "swfd_levels": stages.get("swfd_levels"),
                "parallelism": f"windows sharded in contiguous blocks over {world} GPU(s) x {S} lock-step lane(s) per GPU",
                "lanes_per_gpu": S,
                "labels_sha16": __import__("hashlib").sha256(all_labels.astype(np.int64).tobytes()).hexdigest()[:16],
            },
            "p50_window_latency_ms": float(np.median(lat) * 1e3) if len(lat) else None,
            "stages_ms": stages,
            "roofline": roof,
            "roofline_mfma": roof_gemm,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.kind, args.seed, with_swfd=not args.no_swfd)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    pipe.close()
    if sketch is not None:
        sketch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
