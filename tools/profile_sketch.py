"""rocprofv3 target: a lane-batched sketch at config-2 shapes in steady state (second epoch: MAIN continues the first
window's AUX, duplicates and frozen sketches skipped as in the benchmark).  argv: lanes [rows of the second window]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows2 = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
X = torch.from_numpy(np.stack([np.concatenate([synth.stream_window("blob", 2 * t, W, d, 0)[0],
                                               synth.stream_window("blob", 2 * t + 1, W, d, 0)[0][:rows2]]) for t in range(B)])).cuda()
R = 5500.0
sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
sk.fit_lanes(X[:, :W])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sk.fit_lanes(X[:, W:])
e1.record()
torch.cuda.synchronize()
print("lanes", B, "levels", sk.L, "ms per rotation (steady state):", e0.elapsed_time(e1) / (rows2 / ell))
