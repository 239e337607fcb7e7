# RCCL on ONE rank (STP_FORCE_DP=1): step time per overlap schedule, and which kernels run next to the collective's device kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_dp}; mkdir -p $O
cd $R
run() { label=$1; shift; for rep in 1 2; do env STP_FORCE_DP=1 "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label rep$rep %.3f ms/step' % d['ms_per_step'])" | tee -a $O/dp_single_rank_overlap.txt; done; }
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('no_dp (plain single GPU) %.3f ms/step' % d['ms_per_step'])" | tee -a $O/dp_single_rank_overlap.txt
run overlap_two_phase_70pct STP_DP_OVERLAP=1
run overlap_buckets STP_DP_OVERLAP=buckets
run after_backward STP_DP_OVERLAP=0
run two_phase_nocomm STP_DP_OVERLAP=1 STP_DP_NOCOMM=1
run two_phase_bf16_wire STP_DP_OVERLAP=1 STP_DP_WIRE=bf16
cd /tmp
env STP_FORCE_DP=1 STP_DP_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --sustain 0 > $O/bench_traced.json 2> $O/kt.err
python - <<PY | tee -a $O/dp_single_rank_overlap.txt
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
coll = [r for r in rows if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
print("---- kernel trace (two-phase overlap, one rank): %d collective device kernels in %d kernels" % (len(coll), len(rows)))
names = {}
for c in coll[-6:]:
    ov = {}
    for r in rows:
        if r is c or r["e"] <= c["s"] or r["s"] >= c["e"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        ov[k] = ov.get(k, 0) + (min(r["e"], c["e"]) - max(r["s"], c["s"]))
    print("%-50s %8.1f us, concurrent with: %s" % (c["Kernel_Name"][:50], (c["e"] - c["s"]) / 1e3,
          ", ".join("%s %.1f us" % (k, v / 1e3) for k, v in sorted(ov.items(), key=lambda kv: -kv[1])[:5]) or "nothing"))
PY
rm -rf $O/kt
