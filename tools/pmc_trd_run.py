"""rocprofv3 --pmc target: the direct eigensolver alone on 112 Gram matrices of FD rotation buffers (3 solves)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mused_amd import _lib
from mused_amd.engine import ptr, stream_ptr
from test_gpu_trd import fd_buffers

Gs = fd_buffers("blob", 6)[1:]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 112
L = _lib.lib()
fn = L.mused_debug_trd_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
G = torch.from_numpy(np.stack([Gs[i % len(Gs)] for i in range(batch)])).cuda()
ms = C.c_double()
_lib.check(fn(ptr(G), batch, 2, C.byref(ms), None, None, stream_ptr()))
print("batch", batch, "ms", ms.value)
