"""One batch of 84 sketch-like Gram matrices through the launch-per-round Jacobi (graph solver), for rocprofv3 --pmc."""
import ctypes as C, os, sys
os.environ["MUSED_EIG_QUEUE"] = "0"
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
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 84
G = np.stack([base[i % 28] * (1.0 + 0.01 * (i // 28)) for i in range(batch)])
dG = torch.from_numpy(G).cuda()
ev = torch.empty((batch, n), dtype=torch.float64, device="cuda")
V = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
ms, err = C.c_double(), C.c_int()
rc = fn(dG.data_ptr(), n, batch, 24, 2, ev.data_ptr(), V.data_ptr(), C.byref(ms), C.byref(err), None)
torch.cuda.synchronize()
print(f"batch {batch}: rc={rc} {ms.value:.3f} ms per solve", flush=True)
