import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib, synth
from mused_amd.engine import ptr, stream_ptr
W, d = 10000, 1024
X = torch.from_numpy(synth.stream_window("blob", 0, W, d, 0)[0]).cuda()
norms = torch.empty(W, dtype=torch.float64, device="cuda")
S = torch.empty((W, W), dtype=torch.float64, device="cuda")
_lib.call("mused_pairwise_scores", ptr(X), _lib.F32, W, d, X.stride(0), 0, ptr(norms), ptr(S), stream_ptr())
torch.cuda.synchronize()
print("bitwise symmetric:", bool(torch.equal(S, S.t())), "max |S - S^T|", float((S - S.t()).abs().max()))
