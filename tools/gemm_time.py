"""Times the two batched GEMMs of an FD rotation alone (Gram of the 2l x d buffers, rotate Wc x buffer) through the C ABI:
python tools/gemm_time.py [batch ...]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import _lib
from mused_amd.engine import ptr, stream_ptr

L = _lib.lib()
fn = L.mused_gemm_f64_batched
ell, d = 128, 1024
n2 = 2 * ell
for batch in [int(a) for a in sys.argv[1:]] or [57, 115, 230]:
    buf = torch.randn(batch, n2, d, dtype=torch.float64, device="cuda")
    G = torch.empty(batch, n2, n2, dtype=torch.float64, device="cuda")
    Wc = torch.randn(batch, ell, n2, dtype=torch.float64, device="cuda")
    T = torch.empty(batch, ell, d, dtype=torch.float64, device="cuda")

    def gram():
        _lib.call("mused_gemm_f64_batched", 1, 1, ptr(buf), d, n2 * d, ptr(buf), d, n2 * d, ptr(G), n2, n2 * n2, n2, n2, d, batch, 1.0,
                  stream_ptr())

    def rot():
        _lib.call("mused_gemm_f64_batched", 1, 0, ptr(Wc), n2, ell * n2, ptr(buf), d, n2 * d, ptr(T), d, ell * d, ell, d, n2, batch, 1.0,
                  stream_ptr())

    for name, f, flop in (("gram  (3 tiles of 128 x 128 x 1024 per matrix)", gram, 3 * 2.0 * 128 * 128 * 1024),
                          ("rotate (8 tiles of 128 x 128 x 256 per matrix)", rot, 2.0 * ell * d * n2)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"batch {batch:4d} {name}: {us:8.1f} us  {batch * flop / us / 1e6:6.1f} TFLOP/s", flush=True)
