"""rocprofv3 --pmc target: the blocked direct eigensolver (csrc/trdx.hip) alone on Gram matrices of FD rotation buffers of order
n (2 solves):  python tools/pmc_trdx_run.py [n] [need] [batch]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mused_amd import _lib
from mused_amd.engine import ptr, stream_ptr
from test_gpu_trd import fd_buffers

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
need = int(sys.argv[2]) if len(sys.argv) > 2 else n // 2
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 30
Gs = fd_buffers("blob", 4, ell=n // 2, d=2048)[1:]
L = _lib.lib()
fn = L.mused_debug_trdx_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
G = torch.from_numpy(np.stack([Gs[i % len(Gs)] for i in range(batch)])).cuda()
ms, msa = C.c_double(), C.c_double()
_lib.check(fn(ptr(G), n, need, batch, 1, C.byref(ms), C.byref(msa), None, None, stream_ptr()))
print("order", n, "batch", batch, "ms", ms.value, "tridiagonalisation", msa.value)
