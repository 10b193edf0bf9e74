import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
X = torch.from_numpy(synth.stream_window("blob", 0, W, d, 0)[0]).cuda()
R = float((X.double() ** 2).sum(1).max().item())
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sks = [SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
for sk, st in zip(sks, streams):
    with torch.cuda.stream(st):
        sk.fit(X[:256])
torch.cuda.synchronize()
# one sketch alone
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    sks[0].fit(X)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"single: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
# S sketches on S streams, enqueued round-robin in chunks of 128 rows
t0 = time.perf_counter()
for r0 in range(0, W, 128):
    for sk, st in zip(sks, streams):
        with torch.cuda.stream(st):
            sk.fit(X[r0:r0 + 128])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{S} streams interleaved: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
