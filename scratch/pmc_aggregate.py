"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs -> profiles/<tag>_pmc_traffic.json (bytes per launch and kernel).
usage: python scratch/pmc_aggregate.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)            # argument list
    return name.strip()


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


ft, fc = collect(sys.argv[1], "FETCH_SIZE")
wt, wc = collect(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- python bench.py --steps 2 --warmup 1 "
               "--eager --no-cpu-baseline --no-kernel-profile; counters are KB per dispatch, averaged over all dispatches of a kernel; "
               "fetch_bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section; check: "
               "adam_kernel reads 4 x 97.7 MB); write_bytes = WRITE_SIZE x 1024 (check: adam_kernel writes 3 x 97.7 MB).  Infinity-Cache "
               "hits are counted, so this is L2-miss traffic, an upper bound on HBM traffic.",
       "steps": 3, "kernels": {}}
for k in sorted(set(ft) | set(wt)):
    if not (k.startswith("conv") or k.startswith("bn") or k.startswith("adam") or k.startswith("wgrad") or "kernel" in k and "at::" not in k):
        continue
    out["kernels"][k] = {"dispatches": max(fc.get(k, 0), wc.get(k, 0)),
                         "fetch_bytes_per_launch": int(2 * 1024 * ft[k] / fc[k]) if fc.get(k) else None,
                         "write_bytes_per_launch": int(1024 * wt[k] / wc[k]) if wc.get(k) else None}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["csrc_sha16"] = bench.csrc_fingerprint()     # the build these counters were collected on (bench.py: roofline.counters_stale)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(len(out["kernels"]), "kernels ->", sys.argv[3])
