import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
from mused_amd.engine import WindowEngine
W, d, ell, k, B = 10000, 1024, 128, 50, 4
mode = sys.argv[1]
X = torch.from_numpy(np.stack([synth.stream_window("blob", t, W, d, 0)[0] for t in range(B)])).cuda()
R = float((X[0].double() ** 2).sum(1).max().item())
sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
eng = WindowEngine(W)
if mode == "prio":
    sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
elif mode == "many":
    pool = [torch.cuda.Stream() for _ in range(8)]
    sa, sb = pool[0], pool[5]
else:
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def fa():
    with torch.cuda.stream(sa):
        sk.fit_lanes(X); sk.get_device()
def fb():
    with torch.cuda.stream(sb):
        for p in range(B):
            adj = eng.knn_adjacency(X[p], k); eng.svd_reduce(adj, ell, 0, nnz_cap=W * k)
fa(); fb(); torch.cuda.synchronize()
def timed(fs):
    t0 = time.perf_counter()
    ths = [threading.Thread(target=f) for f in fs]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)
print(f"mode {mode}: sketch alone {timed([fa]):.0f} ms, adjacency+eigenstep alone {timed([fb]):.0f} ms, both {timed([fa, fb]):.0f} ms", flush=True)
