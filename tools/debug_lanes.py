import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
for B in [int(a) for a in sys.argv[1:]]:
    X = torch.from_numpy(np.stack([synth.stream_window("blob", t, W, d, 0)[0] for t in range(B)])).cuda()
    R = float((X[0].double() ** 2).sum(1).max().item())
    sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
    sk.fit_lanes(X[:, :256]); torch.cuda.synchronize()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e0.record(); sk.fit_lanes(X[:, 256:]); e1.record(); sk.get_device(); e2.record(); torch.cuda.synchronize()
    ta, tq = e0.elapsed_time(e1) * W / (W - 256), e1.elapsed_time(e2)
    print(f"lanes {B}: append {ta:.0f} ms ({ta/B:.0f} ms/window), query {tq:.1f} ms, L={sk.L}", flush=True)
    sk.close(); del X
