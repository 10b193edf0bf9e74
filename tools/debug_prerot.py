"""Experiment: rotate every l-row block of the input so that its rows are mutually orthogonal (P' = Q^T P, Q from
eigh(P P^T)) before feeding the sketch.  The sketch is invariant (P'^T P' = P^T P); does the Jacobi need fewer sweeps?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
W, d, ell = 10000, 1024, 128
B = 5
X = torch.from_numpy(np.stack([np.concatenate([synth.stream_window("blob", b * 2 + t, W, d, 0)[0] for t in range(2)]) for b in range(B)])).cuda().double()
R = float((X[0, :W] ** 2).sum(1).max())
def prerot(X):
    Y = X.clone()
    for b in range(B):
        for e in range(2):
            for r0 in range(e * W, (e + 1) * W, ell):
                r1 = min(r0 + ell, (e + 1) * W)
                P = X[b, r0:r1]
                lam, Q = torch.linalg.eigh(P @ P.T)
                if os.environ.get('TORCH_PREROT') == 'desc':
                    Q = Q.flip(1)
                Y[b, r0:r1] = Q.T @ P
    return Y
def run(X, tag):
    sk = SeqBasedSWFD(N=W, R=R, d=d, sketch_dim=ell, lanes=B)
    sk.fit_lanes(X[:, :W].contiguous())
    torch.cuda.synchronize(); sk.profile(True); t0 = time.perf_counter()
    sk.fit_lanes(X[:, W:].contiguous())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, nl, bpl = sk.profile_read()
    full = 16.0 * B * 2 * sk.L * 256 * 256
    _, sig, _ = sk.get_device()
    print(tag, "ms/window/lane %.1f" % (1e3 * dt / B), "eig ms %.0f" % ms, "active sweeps-equivalent %.2f" % (bpl / full * 24), flush=True)
    return sig.cpu().numpy()
s0 = run(X, "plain   ")
s1 = run(prerot(X), "prerot  ") if os.environ.get("TORCH_PREROT") else s0
print("max rel sigma diff", np.abs(s0 - s1).max() / s0.max())
