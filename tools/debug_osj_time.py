import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
L = _lib.lib()
L.mused_debug_osj_time.restype = C.c_int
L.mused_debug_osj_time.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]
rng = np.random.default_rng(0)
for batch in (28, 112, 1):
    B = rng.standard_normal((batch, 256, 1024))
    G = torch.from_numpy(np.einsum("bik,bjk->bij", B, B)).cuda()
    for reps in (150,):
        for v, name in [(0, "full 16 steps"), (7, "4 steps"), (5, "0 steps"), (2, "no rotation maths"), (3, "no barrier"), (4, "no apply")]:
            out = C.c_double()
            rc = L.mused_debug_osj_time(C.c_void_p(G.data_ptr()), batch, v, reps, C.byref(out), None)
            print(f"batch {batch:3d} launches {reps:4d} variant {v} ({name:18s}): {out.value:7.2f} us per launch (16 steps)", flush=True)
