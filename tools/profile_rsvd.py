"""rocprofv3 target: a few eigensteps (randomized SVD) on one config-2 window."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mused_amd import synth
from mused_amd.engine import WindowEngine
W, d, k, ell = 10000, 1024, 50, 128
X = torch.from_numpy(synth.stream_window("blob", 0, W, d, 0)[0]).cuda()
eng = WindowEngine(W)
adj = eng.knn_adjacency(X, k)
for _ in range(int(os.environ.get("REPS", "4"))):
    eng.svd_reduce(adj, ell, 0, nnz_cap=W * k)
torch.cuda.synchronize()
