import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
n, r = 10000, 138
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, r))
P = lambda t: C.c_void_p(t.data_ptr())
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
wi = torch.empty(n + (n + 15) // 16, dtype=torch.int32, device="cuda")
wf = torch.empty(4 * (r + n), dtype=torch.float64, device="cuda")
Y0 = torch.from_numpy(Y).cuda()
for rep in range(3):
    dY = Y0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.call("mused_lu_permute_l", P(dY), n, r, r, P(wi), P(wf), S)
    torch.cuda.synchronize(); print("LU ms", 1e3 * (time.perf_counter() - t0))
