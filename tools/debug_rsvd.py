"""GPU debugging aid: one small window through the device path, stage by stage, against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.engine import WindowEngine
from oracle import mo_oracle as omo

n, d, k, ell, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 500, 64, 50, 16, 0
X, _ = synth.gauss_stream(n, d, 0)
eng = WindowEngine(n)
adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), k)
torch.cuda.synchronize(); print("knn ok", flush=True)
F = omo.create_adjacency_matrix(X, "", k)
m = adj.mask.cpu().numpy().view(np.uint64)
bits = np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
print("adjacency equal:", np.array_equal(bits, F.astype(bool)), flush=True)
emb, sig = eng.svd_reduce(adj, ell, seed, nnz_cap=n * k)
torch.cuda.synchronize(); print("rsvd enqueued+synced", flush=True)
print("status", eng.rsvd_status(), flush=True)
e_ref, s_ref, _ = omo.randomized_svd_reduce(F, ell, seed)
print("sigma dev ", sig.cpu().numpy()[:6])
print("sigma ref ", s_ref[:6])
print("max |emb - ref| / max|ref| =", np.abs(emb.cpu().numpy() - e_ref).max() / np.abs(e_ref).max())
