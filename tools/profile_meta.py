"""Full-size window (W = 10,000) through every modality type of the reference's SED2012 loader on synthetic columns
(mused_amd.synth.metadata_stream / text_stream): per-type adjacency time (host part + device part), fusion, eigenstep.
Checks only what is size independent: degrees, symmetry of the same-user relation, selections against device-side scores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mused_amd import matrix_operations as mo
from mused_amd import synth
from mused_amd.engine import WindowEngine

W, k = 10000, 50
cols, labels = synth.metadata_stream(W, 0, events=8, users=400, vocab=300)
text, _ = synth.text_stream(W, 0, vocab=2000)
eng = WindowEngine(W)
mods = {"location": cols["location"], "time": cols["time"], "username": cols["username"], "tags": cols["tags"], "text": text}
adjs = []
for rep in range(2):
    adjs = []
    for t, m in mods.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a = mo.adjacency_on_device(m, t, k, engine=eng)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        deg = a.degrees()[2].cpu().numpy()
        if rep:
            print(f"{t:9s}: {dt * 1e3:8.1f} ms  max degree {int(deg[0]):4d}  edges {int(deg[1]):8d}", flush=True)
        adjs.append(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
fused = mo.fuse_matrices(adjs)
emb, sigma = mo.svd_reduce_on_device(fused, 50, 0, engine=eng)
torch.cuda.synchronize()
print(f"fuse + eigenstep: {(time.perf_counter() - t0) * 1e3:.1f} ms; sigma[:3] = {sigma[:3].cpu().numpy()}; score workspace allocated: {eng._scores is not None}")
u = adjs[2].to_dense(torch.float64)
assert torch.equal(u, u.t())                      # same-user relation is symmetric
loc_deg = adjs[0].to_dense(torch.float64).sum(1)
valid = ~np.isnan(cols["location"]).any(1)
assert float(loc_deg[torch.from_numpy(valid).cuda()].min()) == k and float(loc_deg[torch.from_numpy(~valid).cuda()].max()) == 0
print("ok")
