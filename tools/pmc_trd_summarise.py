"""FETCH_SIZE / WRITE_SIZE of the direct eigensolver's kernels (csrc/trd.hip) from two rocprofv3 --pmc passes over
tools/pmc_trd_run.py -> the JSON bench.py reads (profiles/rNN_pmc_trd.json):

    python tools/pmc_trd_summarise.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <matrices per launch> [trdx order]

With a fourth argument (the order n of tools/pmc_trdx_run.py) the kernels of the blocked solver (csrc/trdx.hip) are summarised.

Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; FETCH_SIZE
reports half the bytes of wide coalesced streaming reads and is doubled (an upper bound for the kernels whose reads are
8 bytes per lane), WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mused::", "")
        if name.startswith("trd_") or name.startswith("trdx_") or (XN and name.startswith("gemm_f64_kernel")):
            acc[name.split("<")[0]].append(float(r["Counter_Value"]))
    return acc


XN = int(sys.argv[4]) if len(sys.argv) > 4 else 0
fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[3])
algorithmic = {
    "trd_a_kernel": 256 * 257 // 2 * 8 + 254 * 256 * 8,   # lower triangle of G read, Householder vectors written
    "trd_b_kernel": 3 * 256 * 8 + 128 * 8,                 # d, e read; 128 eigenvalues written
    "trd_c_kernel": 4 * 2 * 256 * 8 + 256 * 128 * 8,       # d, e read by 4 workgroups; 128 eigenvectors of T written
    "trd_t_kernel": 254 * 256 * 8 + 16 * 256 * 8,          # Householder vectors read; 16 triangular factors written
    "trd_d_kernel": 254 * 256 * 8 + 256 * 128 * 8 + 16 * 256 * 8 + 256 * 256 * 8,  # V, Z, T read; 256 columns written
}
if XN:
    n = XN
    algorithmic = {
        # symv stream of the lower triangle over the first n - 256 columns + the matrix read once (working copy) + the Householder
        # vectors of those columns and the trailing 256 x 256 block written (the tail goes to trd_a_kernel)
        "trdx_a_kernel": (n ** 3 - 256 ** 3) // 6 * 8 + n * n * 8 + (n - 256) * n * 8 + 256 * 256 * 8,
        "trd_a_kernel": 256 * 257 // 2 * 8 + 254 * 256 * 8,
        "trdx_tail_merge_kernel": 2 * 256 * n * 8,
        "trd_b_kernel": 3 * n * 8,
        "trd_c_kernel": None,
        "trdx_cert_kernel": None,
        "trdx_tfac_kernel": (n // 64) * (n // 64 + 1) // 2 * 64 * 64 * 8 + (n // 64) * 64 * 64 * 8,   # Householder blocks read, T factors written
        "trdx_back_kernel": None,
        "trdx_store_kernel": None,
    }
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --output-format csv -- python3 "
                 f"tools/{'pmc_trdx_run.py ' + str(XN) if XN else 'pmc_trd_run.py'} {batch}, MI355X (launches of {batch} matrices each)",
       "matrices_per_launch": batch, "per_kernel": {}}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fk = sum(fetch.get(k, [0.0])) / max(1, len(fetch.get(k, [])))
    wk = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [])))
    t = (2.0 * fk + wk) * 1024.0 / batch
    tot += t
    out["per_kernel"][k] = {"launches": len(fetch.get(k, [])), "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                            "fetch_bytes_per_matrix_corrected_x2": 2.0 * fk * 1024.0 / batch,
                            "write_bytes_per_matrix": wk * 1024.0 / batch, "traffic_bytes_per_matrix": t,
                            "algorithmic_bytes_per_matrix": algorithmic.get(k)}
out["traffic_bytes_per_matrix"] = tot
out["algorithmic_bytes_per_matrix_chain"] = ((XN ** 3 - 256 ** 3) // 6 * 8 + 2 * XN * XN * 8) if XN else 512 * 1024 + 256 * 1024
if XN:
    out["note"] = ("trdx_a_kernel streams the lower triangle of the panel-start matrix once per column of its first n - 256 columns from L2 / "
                   "Infinity Cache / HBM; the counters sit on the L2's memory side, so re-reads that hit the XCD's L2 are not in them")
else:
  out["note"] = ("chain = trd_a -> trd_b -> trd_c -> trd_t -> trd_d; the Householder vectors (512 KB), the eigenvectors of T (256 KB) "
                 "and the triangular factors travel through global scratch between the kernels, the lower half of the output "
                 "(256 KB of zeros) is written too: the chain's traffic is a few times its end-to-end algorithmic bytes (G read, "
                 "128 columns written) and still far below what HBM delivers in the chain's duration -- it is bound by dependent "
                 "steps, not bytes.")
print(json.dumps(out, indent=1))
