import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
mats = []
B = rng.standard_normal((n, 1024)); mats.append(B @ B.T)                         # well conditioned
B = rng.standard_normal((n, 1024)) * np.logspace(0, -4, n)[:, None]; mats.append(B @ B.T)   # graded (like a shrunk sketch)
B = rng.standard_normal((n, 1024)); B[: n // 2] *= np.linspace(30, 1, n // 2)[:, None]; B[n - 20:] = 0; mats.append(B @ B.T)
B = rng.standard_normal((n, n + 20)); mats.append(B @ B.T)   # kappa(G) ~ 3000: small eigenvalues slow to converge
G = np.stack(mats)
dG = torch.from_numpy(G).cuda()
P = lambda t: C.c_void_p(t.data_ptr())
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for sweeps in [6, 8, 9, 10, 11, 12, 14, 16]:
    ev = torch.empty((len(mats), n), dtype=torch.float64, device="cuda"); V = torch.empty((len(mats), n, n), dtype=torch.float64, device="cuda")
    _lib.call("mused_syevj_batched", P(dG), n, len(mats), sweeps, P(ev), P(V), S)
    torch.cuda.synchronize()
    e, v = ev.cpu().numpy(), V.cpu().numpy()
    out = []
    for b in range(len(mats)):
        sc = np.abs(G[b]).max()
        res = np.abs(G[b] @ v[b] - v[b] * e[b][None, :]).max() / sc
        top = np.argsort(-e[b])[: n // 2]
        vt = v[b][:, top]
        orth = np.abs(vt.T @ vt - np.eye(len(top))).max()
        eref = np.sort(np.linalg.eigvalsh(G[b]))[::-1][: n // 2]
        everr = np.abs(np.sort(e[b])[::-1][: n // 2] - eref).max() / sc
        nzc = e[b] > 1e-9 * sc
        vall = v[b][:, nzc]
        orth_all = np.abs(vall.T @ vall - np.eye(vall.shape[1])).max()
        out.append("res %.0e orthT %.0e orthA %.0e ev %.0e" % (res, orth, orth_all, everr))
    print("sweeps", sweeps, " | ".join(out), flush=True)
