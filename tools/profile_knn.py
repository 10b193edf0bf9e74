"""rocprofv3 target: a few calls of the fused similarity + top-k path on one config-2 window.
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 tools/profile_knn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mused_amd import synth
from mused_amd.engine import WindowEngine

W, d, k = 10000, 1024, 50
X = torch.from_numpy(synth.stream_window("blob", 0, W, d, 0)[0]).cuda()
eng = WindowEngine(W)
for _ in range(int(os.environ.get("REPS", "6"))):
    eng.knn_adjacency(X, k)
torch.cuda.synchronize()
print("fallbacks", eng.knn_fallbacks)
