"""One 9-lane sketch vs two sketches (5 + 4 lanes) on two streams / host threads: time for one window group."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
split = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "9").split("+")]
B = sum(split)
X = torch.from_numpy(np.stack([np.concatenate([synth.stream_window("blob", b * 2 + t, W, d, 0)[0] for t in range(2)]) for b in range(B)])).cuda()
R = float((X[0, :W].double() ** 2).sum(1).max())
sks, sts, lo = [], [], 0
for g in split:
    sks.append((SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=g), lo, lo + g)); lo += g
    sts.append(torch.cuda.Stream(priority=-1))
def run(i, a, b):
    torch.cuda.set_device(0)
    sk, l0, l1 = sks[i]
    with torch.cuda.stream(sts[i]):
        sk.fit_lanes(X[l0:l1, a:b])
        sts[i].synchronize()
def both(a, b):
    ths = [threading.Thread(target=run, args=(i, a, b)) for i in range(len(sks))]
    [t.start() for t in ths]; [t.join() for t in ths]
both(0, 256)
torch.cuda.synchronize(); t0 = time.perf_counter()
both(256, 2 * W)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("split", split, "ms/window/lane", 1e3 * dt / (B * (2 * W - 256) / W))
