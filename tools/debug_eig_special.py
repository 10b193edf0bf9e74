import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(n)
Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
lam = np.concatenate([np.full(n // 4, 7.0), np.full(n // 4, 2.0), np.linspace(1.0, 0.5, n - 2 * (n // 4))])
G = ((Qm * lam) @ Qm.T); G = 0.5 * (G + G.T); G = G[None]
P = lambda t: C.c_void_p(t.data_ptr())
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dG = torch.from_numpy(G).cuda()
for sweeps in [8, 10, 12, 16, 30]:
    ev = torch.empty((1, n), dtype=torch.float64, device="cuda"); V = torch.empty((1, n, n), dtype=torch.float64, device="cuda")
    _lib.call("mused_syevj_batched", P(dG), n, 1, sweeps, P(ev), P(V), S); torch.cuda.synchronize()
    e, v = ev.cpu().numpy()[0], V.cpu().numpy()[0]
    res = np.abs(G[0] @ v - v * e[None, :]).max()
    orth = np.abs(v.T @ v - np.eye(n)).max()
    everr = np.abs(np.sort(e) - np.sort(lam)).max()
    print("cap", sweeps, "res %.1e orth %.1e ev %.1e" % (res, orth, everr), flush=True)
