"""Host-side label stage timing: k-means + Hungarian matching on a (10000, 128) embedding."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import matrix_operations as mo
from concurrent.futures import ThreadPoolExecutor
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
rng = np.random.default_rng(0)
cent = rng.normal(size=(8, 128)) * 3
lab = rng.integers(0, 8, 10000)
embs = [cent[lab] + rng.normal(size=(10000, 128)) for _ in range(8)]
mo.perform_clustering(embs[0], 8, 0)
t0 = time.perf_counter(); cl = [mo.perform_clustering(e, 8, 0) for e in embs]; t1 = time.perf_counter()
print("kmeans sequential ms/window", 1e3 * (t1 - t0) / 8)
t0 = time.perf_counter(); prev = None
for c in cl:
    prev = mo.match_clusters(prev, c, "hungarian", 3)
print("match ms/window", 1e3 * (time.perf_counter() - t0) / 8)
try:
    from threadpoolctl import threadpool_limits
    for nt in (1, 2, 4, 8):
        with threadpool_limits(limits=nt):
            t0 = time.perf_counter(); [mo.perform_clustering(e, 8, 0) for e in embs]; t1 = time.perf_counter()
        print("kmeans threads", nt, "ms/window", 1e3 * (t1 - t0) / 8)
    for nw, nt in ((2, 8), (4, 4), (8, 2), (8, 1), (4, 2)):
        with threadpool_limits(limits=nt):
            with ThreadPoolExecutor(nw) as ex:
                t0 = time.perf_counter(); out = list(ex.map(lambda e: mo.perform_clustering(e, 8, 0), embs)); t1 = time.perf_counter()
        same = all(np.array_equal(a, b) for a, b in zip(out, cl))
        print("pool workers", nw, "threads each", nt, "ms/window", 1e3 * (t1 - t0) / 8, "identical labels", same)
except Exception as e:
    print("threadpoolctl", e)
