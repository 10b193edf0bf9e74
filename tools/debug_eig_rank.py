"""One-sided Jacobi on heavily rank-deficient Gram matrices (rank r of order n)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
n = 256
rng = np.random.default_rng(0)
mats = []
for r in (96, 160, 250):
    B = rng.standard_normal((n, r)) * np.logspace(0, -1.5, r)[None, :]; mats.append(B @ B.T)
    B2 = np.zeros((n, r)); B2[:n // 2 + 20] = B[:n // 2 + 20]; mats.append(B2 @ B2.T)   # plus exact zero rows, like a sketch buffer
G = np.stack(mats)
dG = torch.from_numpy(G).cuda()
P = lambda t: C.c_void_p(t.data_ptr())
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for sweeps in [10, 13, 16, 20, 30]:
    ev = torch.empty((len(mats), n), dtype=torch.float64, device="cuda"); V = torch.empty((len(mats), n, n), dtype=torch.float64, device="cuda")
    _lib.call("mused_syevj_batched", P(dG), n, len(mats), sweeps, P(ev), P(V), S)
    torch.cuda.synchronize()
    e, v = ev.cpu().numpy(), V.cpu().numpy()
    out = []
    for b in range(len(mats)):
        eref = np.sort(np.linalg.eigvalsh(G[b]))[::-1]
        es = np.sort(e[b])[::-1]
        l0 = eref[0]
        big = eref > 1e-9 * l0
        everr = np.abs(es[big] - eref[big]).max() / l0
        nbig_dev = int((es > 1e-10 * l0).sum())
        out.append("ev %.0e n>tol %d/%d" % (everr, nbig_dev, int(big.sum())))
    print("sweeps", sweeps, " | ".join(out), flush=True)
