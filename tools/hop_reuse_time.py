"""Similarity time per window with and without reuse across hopping windows (config-2 shape): JSON on stdout."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.engine import WindowEngine

W, d, k = 10000, 1024, 50
nwin = 6
out = {"W": W, "d": d, "k": k, "windows_timed": nwin, "ratios": {}}
for ratio in (1, 2, 4, 8):
    hop = W // ratio
    rows = np.concatenate([synth.stream_window("blob", t, W, d, 0)[0] for t in range(2 + (hop * (nwin + 1)) // W + 1)])
    X = torch.from_numpy(rows).cuda()
    eng = WindowEngine(W)
    res = {}
    for mode in ("recompute", "reuse"):
        eng._hop.clear()
        def fn(lo):
            eng.begin_window(True)   # flags stay on the device, as in the pipeline (no host read per window)
            if mode == "recompute":
                return eng.knn_adjacency(X[lo:lo + W], k, "l2")
            return eng.knn_adjacency_hop(X[lo:lo + W], k, "l2", key=0, lo=lo)
        fn(0)
        fn(hop)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(2, 2 + nwin):
            fn(t * hop)
        e1.record()
        torch.cuda.synchronize()
        res[mode + "_ms_per_window"] = e0.elapsed_time(e1) / nwin
        res[mode + "_flags"] = int(eng._ovf8.sum().item())
    res["tile_fraction_1_minus_(1-1/ratio)^2"] = 1 - (1 - 1 / ratio) ** 2
    res["measured_fraction"] = res["reuse_ms_per_window"] / res["recompute_ms_per_window"]
    out["ratios"][str(ratio)] = res
    eng.close()
print(json.dumps(out, indent=1))
