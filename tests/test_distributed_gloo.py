"""The N > 1 path on CPU: two gloo ranks shard the windows of a golden stream in contiguous blocks,
all-gather raw k-means labels, replay the Hungarian chain, and must reproduce the reference's
`all_clusters` bit for bit.  The device engine is replaced by the CPU oracle here (tests may use it);
what is under test is mused_amd.distributed + the halo rule for the SWFD sketch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from conftest import load_golden, regen_inputs
    from mused_amd import distributed as md
    from mused_amd import matrix_operations as mo
    from oracle import mo_oracle as omo
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("c1_stream_blob_s0")
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    X = mods[0].astype(np.float64)
    n_win = n // W
    first, b0, b1 = md.windows_with_halo(n_win, world, rank)
    counts = [md.block_partition(n_win, world, r)[1] - md.block_partition(n_win, world, r)[0] for r in range(world)]
    # R from window 0 (rank 0) broadcast to everybody
    R0 = float((X[:W] ** 2).sum(1).max()) if rank == 0 else None
    R = md.broadcast_scalar(R0, 0)
    assert R == float((X[:W] ** 2).sum(1).max())
    raw, sketches = [], []
    sk = OraSWFD(N=W, R=R, d=d, sketch_dim=ell)
    if first > 0:  # start the row counter where the halo window starts
        sk.i = first * W
    for t in range(first, b1):
        Xw = X[t * W : (t + 1) * W]
        sk.fit(Xw)
        if t < b0:
            continue  # halo: primes the MAIN sketches only
        sketches.append(sk.get()[0])
        A = omo.create_adjacency_matrix(Xw, "", k)
        emb, _, _ = omo.randomized_svd_reduce(omo.fuse_matrices([A]), ell, seed)
        raw.append(omo.perform_clustering(emb, len(np.unique(labels[t * W : (t + 1) * W])), seed))
    raw = np.array(raw, dtype=np.int64).reshape(-1, W)
    all_raw = md.gather_raw_labels(raw, counts)
    final = md.replay_label_chain(all_raw, mo.match_clusters)
    np.save(os.path.join(out_dir, f"labels_{rank}.npy"), final)
    np.save(os.path.join(out_dir, f"sketch_{rank}.npy"), np.array(sketches))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_reproduce_the_reference_labels(tmp_path, world):
    """world = 2: even blocks; world = 8 over the 10 windows of the golden stream: UNEVEN blocks (2, 2, 1, 1, ...), i.e.
    the `counts` of gather_raw_labels differ between ranks, as they will on the first real 8-GPU run."""
    from conftest import load_golden, regen_inputs
    from mused_amd import distributed as md
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    if world == 8:
        os.environ.setdefault("OMP_NUM_THREADS", "1")  # 8 ranks on the CPU suite's 8 cores
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("c1_stream_blob_s0")
    for r in range(world):
        got = np.load(tmp_path / f"labels_{r}.npy")
        assert np.array_equal(got.astype(np.int64), g["all_clusters"])  # bit-exact event indices on every rank
    # SWFD: block-with-halo sketches equal the single sequential sketch for every owned window but
    # the very first window of a block that has no halo (rank 0's window 0 is the stream start anyway)
    mods, _, (n, d, W, ell, k, seed) = regen_inputs(g)
    X = mods[0].astype(np.float64)
    seq = OraSWFD(N=W, R=float((X[:W] ** 2).sum(1).max()), d=d, sketch_dim=ell)
    seq_sk = []
    for t in range(n // W):
        seq.fit(X[t * W : (t + 1) * W])
        seq_sk.append(seq.get()[0])
    for r in range(world):
        b0, b1 = md.block_partition(n // W, world, r)
        got = np.load(tmp_path / f"sketch_{r}.npy")
        for j, t in enumerate(range(b0, b1)):
            np.testing.assert_allclose(got[j], seq_sk[t], rtol=0, atol=1e-9 * np.abs(seq_sk[t]).max())


def test_block_partition_covers_everything():
    from mused_amd import distributed as md

    for n_win in (1, 2, 7, 8, 100):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b0, b1 = md.block_partition(n_win, world, r)
                seen.extend(range(b0, b1))
                f, o, e = md.windows_with_halo(n_win, world, r)
                assert f == max(b0 - 1, 0) and o == b0 and e == b1
            assert seen == list(range(n_win))


@pytest.mark.parametrize("n_windows,lanes", [(3, 2), (3, 3), (20, 12), (24, 12), (7, 1), (5, 5), (16, 8)])
def test_lane_schedule_owns_every_window_once_behind_its_halo(n_windows, lanes):
    """The lock-step layout of SwfdmcLanes / bench.py --workload swfdmc: contiguous blocks, every block's first step is the
    window before it (-1 = empty rows at the stream start), every window owned exactly once and in stream order per lane."""
    from mused_amd.distributed import block_partition, lane_schedule

    sched = lane_schedule(n_windows, lanes)
    owned = sorted(i for row in sched for i, own in row if own)
    assert owned == list(range(n_windows))
    for p in range(lanes):
        b0, b1 = block_partition(n_windows, lanes, p)
        col = [row[p] for row in sched]
        assert col[0] == (b0 - 1 if b0 > 0 else -1, False) or b1 == b0   # the halo step
        mine = [i for i, own in col if own]
        assert mine == list(range(b0, b1))                               # in order, contiguous
        fed = [i for i, _ in col]
        assert all(b - a in (0, 1) for a, b in zip(fed, fed[1:]))         # the sketch sees consecutive windows (or a repeat at the end)
