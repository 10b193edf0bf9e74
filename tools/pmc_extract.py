"""Reduce a rocprofv3 --pmc counter_collection.csv to per-kernel averages (small JSON on stdout)."""
import csv, glob, json, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0][:90], r["Counter_Name"], r["Grid_Size"])
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
out = [{"kernel": k[0], "counter": k[1], "grid": int(k[2]), "launches": v[0], "avg": v[1] / v[0]} for k, v in acc.items()]
print(json.dumps(sorted(out, key=lambda x: -x["avg"])[:40], indent=1))
