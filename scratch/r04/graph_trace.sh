cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_trace}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktg -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-feed --no-kernel-profile > $O/bench_profiled_graph_only.json 2> $O/ktg.err
cp $(find $O/ktg -name "*kernel_stats.csv" | head -1) $O/kernel_stats_graph_only.csv
python - <<PY > $O/graph_gaps.txt
import csv,glob
f=glob.glob("$O/ktg/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "augment_kernel" in r["Kernel_Name"]]
idx=idx[-10:]
seg=rows[idx[0]:]
# wall clock per step from augmentation kernel to augmentation kernel (9 intervals; kernels launched after the run do not count)
wall=(int(rows[idx[-1]]["Start_Timestamp"])-int(rows[idx[0]]["Start_Timestamp"]))/(len(idx)-1)
inner=rows[idx[0]:idx[-1]]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in inner)/(len(idx)-1)
print("last %d steps: kernels/step %.1f  wall/step %.1f us  sum of kernel durations/step %.1f us  gaps/step %.1f us"%(len(idx),len(inner)/(len(idx)-1),wall/1e3,busy/1e3,(wall-busy)/1e3))
fam={}
for r in seg:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    fam[k]=fam.get(k,[0,0]); fam[k][0]+=1; fam[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in sorted(fam.items(),key=lambda kv:-kv[1][1]): print("%-70s %6.1f launches/step %9.1f us/step"%(k[:70],v[0]/len(idx),v[1]/1e3/len(idx)))
PY
rm -rf $O/ktg
