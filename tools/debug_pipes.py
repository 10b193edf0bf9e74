import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.pipeline import StreamPipeline
from mused_amd.swfd import SeqBasedSWFD
W, d, ell, k = 10000, 1024, 128, 50
S = int(sys.argv[1]); mode = sys.argv[2]   # mode: full | nosketch | nolabels
wins = [synth.stream_window("blob", t, W, d, 0) for t in range(3)]
X = [torch.from_numpy(x).cuda() for x, _ in wins]; lab = [l for _, l in wins]
R = float((X[0].double() ** 2).sum(1).max().item())
pipes = []
for p in range(S):
    pp = StreamPipeline(W, ell, k, 0, "sSVDMC", feature_sketch=(mode != "nosketch"), async_labels=True, stream=torch.cuda.Stream())
    if mode != "nosketch":
        pp.fswfd = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell)
    if mode == "nolabels":
        pp._labels = lambda *a, **k: None
    pipes.append(pp)
for pp in pipes:
    pp.process_window([X[0]], lab[0])
for pp in pipes: pp.flush()
torch.cuda.synchronize()
import threading
def drive(pp):
    for t in (1, 2):
        pp.process_window([X[t]], lab[t])
t0 = time.perf_counter()
ths = [threading.Thread(target=drive, args=(pp,)) for pp in pipes]
for th in ths: th.start()
for th in ths: th.join()
t1 = time.perf_counter()
for pp in pipes: pp.flush()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"S={S} mode={mode}: enqueue {1e3*(t1-t0):.0f} ms, total {1e3*(t2-t0):.0f} ms for {2*S} windows -> {1e3*(t2-t0)/(2*S):.0f} ms/window", flush=True)
