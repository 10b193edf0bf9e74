"""The N > 1 path with the DEVICE engine under every rank: two gloo ranks share cuda:0 (the rehearsal the one-GPU box
allows; the collectives are those of the RCCL run, on the CPU backend), shard the windows of a golden stream in contiguous
blocks, run adjacency -> eigenstep -> k-means through StreamPipeline and the feature-row sketch through the HIP
SeqBasedSWFD (halo window first, rank hand-over through begin_epoch), all-gather raw labels, replay the Hungarian chain:
event labels bit-identical to the reference's run on every rank, sketches equal to the sequential sketch (oracle)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

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
    import torch
    import torch.distributed as dist

    from conftest import load_golden, regen_inputs
    from mused_amd import distributed as md
    from mused_amd import matrix_operations as mo
    from mused_amd.pipeline import StreamPipeline
    from mused_amd.swfd import SeqBasedSWFD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    g = load_golden("c1_stream_blob_s0")
    mods, labels, (n, d, W, ell, k, seed) = regen_inputs(g)
    X = torch.from_numpy(mods[0].astype(np.float64)).cuda()
    n_win = n // W
    first, b0, b1 = md.windows_with_halo(n_win, world, rank)
    counts = [md.block_partition(n_win, world, r)[1] - md.block_partition(n_win, world, r)[0] for r in range(world)]
    R0 = float((X[:W] ** 2).sum(1).max().item()) if rank == 0 else None
    R = md.broadcast_scalar(R0, 0)
    sketches = []
    sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell)
    if first > 0:
        sk.begin_epoch(first * W)  # the row counter starts where the halo window starts
    with StreamPipeline(W, ell, k, seed, "sSVDMC", modality_types=[""]) as pipe:
        for t in range(first, b1):
            rows = X[t * W:(t + 1) * W]
            sk.fit(rows)
            if t < b0:
                continue  # halo: primes the sketch only
            sketches.append(np.asarray(sk.get()[0]))
            pipe.process_window([rows], labels[t * W:(t + 1) * W], trigger=(t + 1) * W - 1)
        pipe.flush()
        by_trigger = {tr["trigger"]: tr["raw"] for tr in pipe.trace}
    sk.close()
    raw = np.array([by_trigger[(t + 1) * W - 1] for t in range(b0, b1)], dtype=np.int64).reshape(-1, W)
    all_raw = md.gather_raw_labels(raw, counts)
    final = md.replay_label_chain(all_raw, mo.match_clusters)
    np.save(os.path.join(out_dir, f"labels_{rank}.npy"), final)
    np.save(os.path.join(out_dir, f"sketch_{rank}.npy"), np.array(sketches))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_device_reproduce_the_reference_labels(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    from conftest import load_golden, regen_inputs
    from mused_amd import distributed as md
    from oracle.swfd_oracle import SeqBasedSWFD as OraSWFD

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("c1_stream_blob_s0")
    for r in range(world):
        got = np.load(tmp_path / f"labels_{r}.npy")
        assert np.array_equal(got.astype(np.int64), g["all_clusters"])
    mods, _, (n, d, W, ell, k, seed) = regen_inputs(g)
    X = mods[0].astype(np.float64)
    seq = OraSWFD(N=W, R=float((X[:W] ** 2).sum(1).max()), d=d, sketch_dim=ell)
    seq_sk = []
    for t in range(n // W):
        seq.fit(X[t * W:(t + 1) * W])
        seq_sk.append(seq.get()[0])
    for r in range(world):
        b0, b1 = md.block_partition(n // W, world, r)
        got = np.load(tmp_path / f"sketch_{r}.npy")
        for j, t in enumerate(range(b0, b1)):
            # the sketch is a basis up to row signs / rotations inside equal singular values: compare the Gram matrices
            G0, G1 = got[j].T @ got[j], seq_sk[t].T @ seq_sk[t]
            np.testing.assert_allclose(G0, G1, rtol=0, atol=1e-8 * np.abs(G1).max())
