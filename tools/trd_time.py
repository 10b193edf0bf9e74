"""Times the direct eigensolver (csrc/trd.hip) alone on Gram matrices of FD rotation buffers: python tools/trd_time.py [batch...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mused_amd import _lib
from mused_amd.engine import ptr, stream_ptr
from test_gpu_trd import fd_buffers

Gs = fd_buffers("blob", 6)[1:]  # steady-state buffers (kept rows + a new block)
L = _lib.lib()
fn = L.mused_debug_trd_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
for batch in [int(a) for a in sys.argv[1:]] or [1, 28, 112, 224, 256, 280, 512]:
    G = torch.from_numpy(np.stack([Gs[i % len(Gs)] for i in range(batch)])).cuda()
    ms = C.c_double()
    done = (C.c_int * batch)()
    clk = torch.zeros(batch, 16, dtype=torch.int64, device="cuda")
    _lib.check(fn(ptr(G), batch, 5, C.byref(ms), done, ptr(clk), stream_ptr()))
    c = clk.cpu().numpy().astype(np.float64)
    print(f"batch {batch:4d}: {ms.value:8.3f} ms per solve   ({sum(done)} of {batch} certified)   (per-kernel times: rocprofv3 "
          f"--kernel-trace --stats on this script)", flush=True)
    if c[:, 8:14].any():
        st = c[:, 8:14].mean(axis=0)
        print("      phase A step parts (clock64 ticks, summed over the steps): extract+barrier %.0f | vector+barrier %.0f | symv+reduce %.0f | "
              "barrier %.0f | w+barrier %.0f | update %.0f" % tuple(st), flush=True)
