import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from mused_amd import _lib as L
P = lambda t: C.c_void_p(t.data_ptr())
for n in (320, 448, 512, 256):
    rng = np.random.default_rng(n)
    mats = [np.zeros((n, n)), np.diag(np.linspace(5.0, 0.0, n))]
    u = rng.standard_normal(n); mats.append(np.outer(u, u))
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.full(n // 4, 7.0), np.full(n // 4, 2.0), np.linspace(1.0, 0.5, n - 2 * (n // 4))])
    mats.append((Qm * lam) @ Qm.T)
    B = rng.standard_normal((n, n // 3)); mats.append(B @ B.T)
    G = np.stack([0.5 * (m + m.T) for m in mats])
    dG = torch.from_numpy(G).cuda()
    outs = []
    for rep in range(4):
        ev = torch.empty((len(mats), n), dtype=torch.float64, device="cuda")
        V = torch.empty((len(mats), n, n), dtype=torch.float64, device="cuda")
        L.call("mused_syevj_batched", P(dG), n, len(mats), 30, P(ev), P(V), None)
        torch.cuda.synchronize()
        evn, Vn = ev.cpu().numpy(), V.cpu().numpy()
        errs = []
        for b in range(len(mats)):
            scale = max(np.abs(G[b]).max(), 1e-300)
            nz = evn[b] > 1e-9 * scale
            Vb = Vn[b][:, nz]
            errs.append(float(np.abs(Vb.T @ Vb - np.eye(int(nz.sum()))).max()) if nz.any() else 0.0)
        outs.append((evn.copy(), Vn.copy()))
        print(n, rep, ["%.2e" % e for e in errs], "identical to rep0:", np.array_equal(outs[0][1], Vn), flush=True)
