# kernel-trace of the step graph: sum of kernel durations vs the step time (the difference = inter-kernel gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > $O/bench_profiled.json 2> $O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python - <<PY
import csv,glob
f=glob.glob("$O/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 10 steps: find augment kernel starts as step delimiters
idx=[i for i,r in enumerate(rows) if "augment_kernel" in r["Kernel_Name"]]
idx=idx[-10:]
seg=rows[idx[0]:]
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)
print("last %d steps: kernels/step %.1f  wall/step %.1f us  sum of kernel durations/step %.1f us  gaps/step %.1f us"%(len(idx),len(seg)/len(idx),(t1-t0)/1e3/len(idx),busy/1e3/len(idx),((t1-t0)-busy)/1e3/len(idx)))
PY
rm -rf $O/kt
