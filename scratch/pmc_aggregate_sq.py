"""rocprofv3 SQ counter pass -> profiles/<tag>_pmc_sq.json: per kernel MFMA pipe utilisation, LDS activity / bank conflicts, wave stalls.
usage: python scratch/pmc_aggregate_sq.py <counter_collection.csv> <kernel_trace.csv> <out.json>

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz): busy cycles are summed over the chip's SIMDs
(check: one v_mfma_f32_16x16x32_bf16 = 16, one 32x32x16 = 32 busy cycles, so the sum = MFMA count x passes), the duration is the
dispatch's End - Start from the kernel trace of the SAME run; normalised to the PEAK clock, so it is directly the fraction of the
2.5 PFLOP/s figure the MFMA pipe was busy (a kernel at the DVFS clock of ~2.0 GHz cannot exceed ~0.83).
lds_util = SQ_LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz); bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in s_waitcnt / s_barrier), issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


cnt = defaultdict(lambda: defaultdict(float))
nd = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        k = short(r["Kernel_Name"])
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nd[k][r["Counter_Name"]] += 1
dur, dn = defaultdict(float), defaultdict(int)
with open(sys.argv[2], newline="") as f:
    for r in csv.DictReader(f):
        k = short(r["Kernel_Name"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        dn[k] += 1
out = {"_how": __doc__, "kernels": {}}
for k in sorted(cnt):
    if "at::" in k or not dn.get(k):
        continue
    c = {n: cnt[k][n] / nd[k][n] for n in cnt[k]}
    d_ns = dur[k] / dn[k]
    cyc = d_ns * 2.4
    e = {"dispatches": dn[k], "avg_duration_us": round(d_ns / 1e3, 2)}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        e["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
    if "SQ_LDS_IDX_ACTIVE" in c:
        e["lds_util"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 4)
        e["bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        e["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        e["issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        e["active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
    e["counters"] = {n: int(v) for n, v in c.items()}
    out["kernels"][k] = e
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["csrc_sha16"] = bench.csrc_fingerprint()     # the build these counters were collected on (bench.py: roofline.counters_stale)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(len(out["kernels"]), "kernels ->", sys.argv[3])
