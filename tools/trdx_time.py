"""Times the blocked direct eigensolver of orders 320 .. 1024 (csrc/trdx.hip) alone on Gram matrices of FD rotation buffers,
beside the one-sided Jacobi of the same order:  python tools/trdx_time.py [n:need:batch ...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mused_amd import _lib
from mused_amd.engine import ptr, stream_ptr
from test_gpu_trd import fd_buffers

L = _lib.lib()
fn = L.mused_debug_trdx_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
fj = L.mused_debug_eig_time
fj.restype = C.c_int
fj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(512, 256, 30), (512, 256, 120), (384, 128, 7), (384, 128, 20), (512, 128, 20), (768, 256, 8), (1024, 256, 8)]
bufs = {}
for n, need, batch in cases:
    if n not in bufs:
        bufs[n] = fd_buffers("blob", 5, ell=n // 2, d=1024 if n < 512 else 2048)[1:]
    Gs = bufs[n]
    G = torch.from_numpy(np.stack([Gs[i % len(Gs)] for i in range(batch)])).cuda()
    ms, msa = C.c_double(), C.c_double()
    done = (C.c_int * batch)()
    prof = torch.zeros(batch, 4, dtype=torch.int64, device="cuda")
    _lib.check(fn(ptr(G), n, need, batch, 3, C.byref(ms), C.byref(msa), done, ptr(prof) if os.environ.get("TRDX_PROF") else None, stream_ptr()))
    line = f"order {n} top {need} batch {batch:4d}: direct {ms.value:8.3f} ms per solve (blocked tridiagonalisation of the first {n - 256} columns {msa.value:8.3f}), {sum(done)} of {batch} certified"
    if "--jacobi" in os.environ.get("TRDX_TIME", "--jacobi"):
        ev = torch.zeros(batch, n, dtype=torch.float64, device="cuda")
        V = torch.zeros(batch, n, n, dtype=torch.float64, device="cuda")
        mj = C.c_double()
        err = C.c_int()
        _lib.check(fj(ptr(G), n, batch, 24, 2, ptr(ev), ptr(V), C.byref(mj), C.byref(err), stream_ptr()))
        line += f" | one-sided Jacobi {mj.value:8.3f} ms"
    print(line, flush=True)
    if os.environ.get("TRDX_PROF"):
        pr = prof.cpu().numpy().astype(np.float64).mean(axis=0)
        pr = 100.0 * pr / pr.sum()                                         # s_memtime ticks of thread 0 -> shares of the kernel
        print("      blocked kernel A, shares of its time (thread 0): column + Householder %.0f %% | symv %.0f %% | reductions + w %.0f %% | panel update %.0f %%" % tuple(pr), flush=True)
