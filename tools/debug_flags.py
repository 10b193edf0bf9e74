import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.engine import WindowEngine
W, d, k, ell, seed = 500, 64, 50, 16, 0
X, _ = synth.gauss_stream(W * 30, d, 0)
eng = WindowEngine(2048)
bad = 0
for w in range(30):
    Xw = X[w*W:(w+1)*W]
    adj = eng.knn_adjacency(torch.from_numpy(Xw).cuda(), k)
    fused = eng.fuse([adj])
    emb, sig = eng.svd_reduce(fused, ell, seed, nnz_cap=W * k)
    st = eng.rsvd_status()
    if st[0] != 0:
        bad += 1
        print("window", w, "flags", hex(st[0] & 0xffffffff), flush=True)
print("bad windows:", bad, "env", {k: v for k, v in os.environ.items() if k.startswith("MUSED")})
