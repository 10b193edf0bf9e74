"""Small rocprofv3 --pmc target: a few FD rotations of a lane-batched sketch at config-2 shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X = torch.from_numpy(np.stack([synth.stream_window("blob", t, W, d, 0)[0][:512] for t in range(B)])).cuda()
R = 5500.0
sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
sk.fit_lanes(X)      # 4 rotations
torch.cuda.synchronize()
print("done", sk.L)
