"""Persistent work-queue Jacobi (MUSED_EIG_QUEUE=1) vs the launch-per-round graph: identical results, timings.
Sketch-like Gram matrices of order 256 (kept rows orthogonal + a fresh block), batches of 28 / 84 / 280."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mused_amd import _lib

L = _lib.lib()
fn = L.mused_debug_eig_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
rng = np.random.default_rng(0)
n, d = 256, 1024
def gram():
    K = np.linalg.qr(rng.standard_normal((d, 127)))[0].T * np.sort(rng.uniform(30, 3000, 127))[::-1, None] ** 0.5
    P = rng.standard_normal((128, d))
    w, V = np.linalg.eigh(P @ P.T)
    B = np.vstack([K, V[:, ::-1].T @ P, np.zeros((1, d))])
    return B @ B.T
base = [gram() for _ in range(28)]
for batch in (28, 84, 280):
    G = np.stack([base[i % 28] * (1.0 + 0.01 * (i // 28)) for i in range(batch)])
    dG = torch.from_numpy(G).cuda()
    res = {}
    for mode in ("0", "1"):
        os.environ["MUSED_EIG_QUEUE"] = mode
        ev = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        V = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        ms, err = C.c_double(), C.c_int()
        rc = fn(dG.data_ptr(), n, batch, 24, 5, ev.data_ptr(), V.data_ptr(), C.byref(ms), C.byref(err), None)
        torch.cuda.synchronize()
        res[mode] = (ev.cpu().numpy(), V.cpu().numpy(), ms.value, err.value, rc)
        print(f"batch {batch} queue={mode}: rc={rc} {ms.value:.3f} ms per solve, info={err.value}", flush=True)
    same = np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    ref = np.linalg.eigvalsh(G[0])[::-1]
    got = np.sort(res["1"][0][0])[::-1]
    print(f"  bit-identical: {same}; max rel err of eigenvalues vs LAPACK: {np.abs(got - ref).max() / ref[0]:.2e}", flush=True)
