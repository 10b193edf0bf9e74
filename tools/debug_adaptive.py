"""Sketch time per window and result difference: fixed sweeps vs MUSED_EIG_ADAPTIVE=1 (run once per mode)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nwin = 2
X = torch.from_numpy(np.stack([np.concatenate([synth.stream_window("blob", b * nwin + t, W, d, 0)[0] for t in range(nwin)]) for b in range(B)])).cuda()
R = float((X[0, :W].double() ** 2).sum(1).max())
SW = int(os.environ.get('SW', '0'))
sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B, sweeps=SW)
sk.fit_lanes(X[:, :256].contiguous())
torch.cuda.synchronize()
sk.profile(True)
t0 = time.perf_counter()
sk.fit_lanes(X[:, 256:].contiguous())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, nl, bpl = sk.profile_read()
full = 16.0 * B * 2 * sk.L * 256 * 256
print('eig ms total', ms, 'launches', nl, 'us/launch', 1e3 * ms / max(nl, 1), 'active fraction', bpl / full, 'GB/s', bpl / (1e-3 * ms / max(nl, 1)) / 1e9)
Bm, sig, info = sk.get_device()
sig = sig.cpu().numpy()
print("sweeps", SW, "mode", os.environ.get("MUSED_EIG_ADAPTIVE", "0"), "L", sk.L, "ms/window/lane", 1e3 * dt / (B * (nwin * W - 256) / W))
np.save(sys.argv[2], sig)
if len(sys.argv) > 3:
    ref = np.load(sys.argv[3])
    print("max rel sigma diff vs", sys.argv[3], np.abs(sig - ref).max() / ref.max())
