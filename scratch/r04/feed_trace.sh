# where does the fed step lose time?  kernel + memory-copy trace of a fed run, and A/Bs of the copy path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_feed}; mkdir -p $O
cd $R
ab() { label=$1; shift; env "$@" python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label resident %.3f fed %.3f ms' % (d['ms_per_step'], d['ms_per_step_with_feed']))" | tee -a $O/feed_ab.txt; }
ab default
ab sdma_off HSA_ENABLE_SDMA=0
ab default_again
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o tr -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-profile --sustain 0 > $O/bench_traced.json 2> $O/tr.err
python - <<PY | tee $O/feed_trace.txt
import csv, glob
kt = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)
mc = glob.glob("$O/tr/**/*memory_copy_trace.csv", recursive=True)
print("files", kt, mc)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    print("memory copies:", len(rows), rows[0].keys() if rows else "")
    big = [r for r in rows if int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)) > 50000]
    for r in big[-12:]:
        print({k: r[k] for k in r if k in ("Direction", "Start_Timestamp", "End_Timestamp", "Source_Agent_Id", "Destination_Agent_Id")}, "dur us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if kt:
    rows = list(csv.DictReader(open(kt[0])))
    cp = [r for r in rows if "copy" in r["Kernel_Name"].lower()]
    print("copy-like kernels:", len(cp))
    for r in cp[-8:]:
        print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
rm -rf $O/tr
