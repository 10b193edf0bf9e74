"""Per-stream busy shares and kernel-time breakdown from a rocprofv3 --kernel-trace csv (compacted: name,queue,stream,start,end,wg,grid):

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python3 bench.py --no-cpu-baseline --no-single-lane
    python tools/trace_streams.py <compact csv[.gz]> [t0_ms t1_ms]      # window of the timeline to break down

The compact file is what tools/scratch/trace_c2.sh writes (kernel names cut to 40 characters; they may contain commas: the six
numeric fields are split from the right)."""
import collections
import gzip
import sys

import numpy as np

path = sys.argv[1]
op = gzip.open if path.endswith(".gz") else open
rows = [line.rstrip("\n").rsplit(",", 6) for i, line in enumerate(op(path, "rt")) if i > 0]
names = np.array([r[0] for r in rows])
s = np.array([r[2] for r in rows])
st = np.array([int(r[3]) for r in rows], dtype=np.float64)
en = np.array([int(r[4]) for r in rows], dtype=np.float64)
t0 = st.min()
st, en = (st - t0) / 1e6, (en - t0) / 1e6
streams = [k for k, _ in collections.Counter(s).most_common(8)]
print(f"{len(rows)} kernels over {en.max():.0f} ms; busy share of every stream per 50 ms (% of the bin with a kernel of that stream executing)")
print("t(ms)  " + " ".join(f"{x:>5}" for x in streams))
for b0 in np.arange(0, en.max(), 50):
    b1 = b0 + 50
    out = []
    for ss in streams:
        m = (s == ss) & (en > b0) & (st < b1)
        out.append((np.minimum(en[m], b1) - np.maximum(st[m], b0)).sum() / 50 * 100)
    print(f"{int(b0):5d}  " + " ".join(f"{x:5.0f}" for x in out))
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    for ss in streams[:4]:
        m = (s == ss) & (st >= a) & (en < b)
        tot = collections.Counter()
        for n, d in zip(names[m], (en - st)[m]):
            tot[n[:36]] += d
        print(f"stream {ss}: kernel ms in [{a:.0f}, {b:.0f}) ms ({sum(tot.values()):.0f} of {b - a:.0f} busy)")
        for k, v in tot.most_common(12):
            print(f"    {k:38s} {v:7.1f}")
