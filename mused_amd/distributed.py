"""Sharding the window stream over the GPUs of one node.

Unit of parallelism: the tumbling window (main.py:32 with step_window_ratio = 1).  Adjacency,
fusion, eigenstep and k-means of window t depend on that window's rows only.  Two pieces of state
cross windows:

  * the SWFD sketch.  With N = W the MAIN sketches of window t are the AUX sketches of window t-1
    continued over window t, and the AUX sketches of any window start from empty.  So a rank that
    owns the CONTIGUOUS block of windows [b0, b1) reproduces the sequential sketch for every window
    of its block except the first by simply starting one window early: window b0 - 1 is its halo
    (its own AUX pass primes MAIN for b0).  No sketch state crosses ranks.  R (main.py:61) is fixed
    from window 0: one broadcast of a double from rank 0.
  * the label chain match_clusters(prev, new) (main.py:105-118), sequential over windows.  Each rank
    all-gathers its RAW k-means labels (W ints per window); the chain is replayed on every rank
    (milliseconds of host work), so the final labels equal the single-process run bit for bit.

Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).
The only collectives are that broadcast and that all-gather -- the data path has none.
"""
from __future__ import annotations

import numpy as np


def block_partition(n_windows: int, world_size: int, rank: int):
    """Contiguous block [b0, b1) of windows owned by `rank` (sizes differ by at most one)."""
    q, r = divmod(n_windows, world_size)
    b0 = rank * q + min(rank, r)
    return b0, b0 + q + (1 if rank < r else 0)


def windows_with_halo(n_windows: int, world_size: int, rank: int):
    """(first window to feed the sketch, first owned window, end).  The halo window b0 - 1 primes
    the SWFD MAIN sketches; it produces no labels."""
    b0, b1 = block_partition(n_windows, world_size, rank)
    return (max(b0 - 1, 0), b0, b1)


def lane_schedule(n_windows: int, lanes: int):
    """Lock-step schedule of `lanes` contiguous blocks of a stream of n_windows windows, each preceded by its halo window:
    a list over lock-steps of per-lane (window index, owned) pairs.  Step 0 is the halo step (window b0 - 1 of every block;
    -1 = "a window of empty rows": the block that starts the stream); a lane whose block is shorter than the longest one
    repeats its last window with owned = False (its results are discarded).  Every window is owned exactly once."""
    blocks = [block_partition(n_windows, lanes, p) for p in range(lanes)]
    steps = 1 + max(b1 - b0 for b0, b1 in blocks)
    out = []
    for t in range(steps):
        row = []
        for b0, b1 in blocks:
            idx = b0 - 1 + t
            own = t >= 1 and idx < b1
            row.append((min(idx, b1 - 1), own))
        out.append(row)
    return out


def broadcast_scalar(value, src: int = 0, device=None) -> float:
    """R of main.py:61 from the owner of window 0 to every rank."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value) if value is not None else 0.0], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return float(t.item())


def gather_raw_labels(raw_local: np.ndarray, counts, device=None) -> np.ndarray:
    """All-gather per-window raw labels.  raw_local: (n_local_windows, W) int64; counts[r] = number of
    windows rank r owns.  Returns (n_windows, W) in window order on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(raw_local)
    W = raw_local.shape[1] if raw_local.size else 0
    world = dist.get_world_size()
    if W == 0:  # a rank may own no window; learn W from the others
        wt = torch.tensor([0], dtype=torch.int64, device=device)
    else:
        wt = torch.tensor([W], dtype=torch.int64, device=device)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    W = int(wt.item())
    mx = max(counts)
    pad = torch.zeros((mx, W), dtype=torch.int64, device=device)
    if raw_local.size:
        pad[: raw_local.shape[0]] = torch.from_numpy(np.ascontiguousarray(raw_local)).to(pad.device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return np.concatenate([p[: counts[r]].cpu().numpy() for r, p in enumerate(parts)], axis=0)


def replay_label_chain(raw_windows: np.ndarray, match_fn) -> np.ndarray:
    """main.py:105-119 over gathered raw labels: matched_t = match(matched_{t-1}, raw_t)."""
    prev = None
    out = []
    for raw in raw_windows:
        matched = match_fn(prev, raw, method="hungarian", min_overlap=3)
        if matched is None or len(matched) == 0:
            matched = np.full(len(raw), 0)
        prev = matched
        out.extend(matched)
    return np.array(out)
