import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth, _lib
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
X = torch.from_numpy(np.concatenate([synth.stream_window("blob", t, W, d, 0)[0] for t in range(2)])).cuda()
R = float((X[:W].double() ** 2).sum(1).max())
sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell)
L = _lib.lib()
S = 2 * sk.L
for upto in (1280, 5120, 10000, 12800, 19200):
    sk.fit(X[sk.rows_seen:upto])
    rep = (C.c_int * S)(); meta = (C.c_int * (4 * S))()
    rc = L.mused_swfd_debug_state(sk._h, rep, meta)
    rep = np.array(rep); meta = np.array(meta).reshape(S, 4)
    print(upto, "rc", rc, "followers", int((rep != np.arange(S)).sum()), "rep", rep.tolist(), "ndump", meta[:, 3].tolist(), flush=True)
