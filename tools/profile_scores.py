"""Runs only the similarity stage (row norms + fp64 MFMA score GEMM + selection) of one config-2 window a
few times: small target for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE of the dominant kernel)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth, _lib
from mused_amd.engine import WindowEngine
W, d, k = 10000, 1024, 50
X = torch.from_numpy(synth.stream_window("blob", 0, W, d, 0)[0]).cuda()
eng = WindowEngine(W)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.knn_adjacency(X, k)
torch.cuda.synchronize()
print("done")
