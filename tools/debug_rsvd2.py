import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.engine import WindowEngine
from oracle import mo_oracle as omo
n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, d, k, ell, seed = 500, 64, 50, 16, 0
X, _ = synth.gauss_stream(1500, d, 0)
eng = WindowEngine(n_max)
for w in range(3):
    t0 = time.time()
    Xw = X[w*W:(w+1)*W]
    adj = eng.knn_adjacency(torch.from_numpy(Xw).cuda(), k)
    fused = eng.fuse([adj])
    R = eng.max_row_sq_norm(fused)
    emb, sig = eng.svd_reduce(fused, ell, seed, nnz_cap=W * k)
    st = eng.rsvd_status()
    torch.cuda.synchronize()
    F = omo.create_adjacency_matrix(Xw, "", k)
    e_ref, s_ref, _ = omo.randomized_svd_reduce(F, ell, seed)
    print(w, "R", R, "status", st, "dt %.2f" % (time.time() - t0),
          "sig err", np.abs(sig.cpu().numpy() - s_ref).max(),
          "emb err", np.abs(emb.cpu().numpy() - e_ref).max() / np.abs(e_ref).max(), flush=True)
