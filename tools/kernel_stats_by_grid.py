"""Per-kernel, per-grid-size duration statistics from a rocprofv3 --kernel-trace CSV (the --stats table averages a kernel over
all its launches; the lock-step sketch launches of bench.py differ by grid: timed region / halo windows / single lane):

    python tools/kernel_stats_by_grid.py <dir with *kernel_trace.csv> [name filter] > small.csv"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0]
    if flt and flt not in name:
        continue
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    acc[(name, grid // max(1, wg))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,workgroups,calls,avg_ns,min_ns,max_ns")
for (name, wgs), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{name}",{wgs},{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)}')
