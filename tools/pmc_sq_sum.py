import csv, glob, sys, collections, json
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    flt = sys.argv[2] if len(sys.argv) > 2 else "osj"
    if flt in k:
        n = max(cnt[(k, c)] for c in v)
        print(k, "launches", n, json.dumps({c: round(x / n, 1) for c, x in v.items()}))
