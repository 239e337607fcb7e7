# round-6 profile set for ONE BASELINE config: bench line, rocprofv3 kernel stats (graph-only), PMC traffic (FETCH / WRITE: separate passes), SQ counters,
# GEMM floor table.   usage (on the GPU box): bash scratch/r06/prof_r06.sh <tag> <config 1|3|4> [full]     -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r06z}
CFG=${2:-1}
FULL=${3:-short}
SUF=""; [ "$CFG" != "1" ] && SUF="_config$CFG"
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
Q="--config $CFG --no-cpu-baseline --sustain 0 --no-feed --no-calibration"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch$SUF -o f -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_fetch$SUF.json 2> $O/pmc_fetch$SUF.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write$SUF -o w -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_write$SUF.json 2> $O/pmc_write$SUF.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq$SUF -o q -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_sq$SUF.json 2> $O/pmc_sq$SUF.err
cd $R
python scratch/pmc_aggregate.py $(find $O/pmc_fetch$SUF -name "*counter_collection.csv" | head -1) $(find $O/pmc_write$SUF -name "*counter_collection.csv" | head -1) $O/pmc_traffic$SUF.json
python scratch/pmc_aggregate_sq.py $(find $O/pmc_sq$SUF -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq$SUF -name "*kernel_trace.csv" | head -1) $O/pmc_sq$SUF.json
# the counter files of THIS build in place before the bench line that reads them
mkdir -p $R/profiles
cp $O/pmc_traffic$SUF.json $R/profiles/${T}_pmc_traffic$SUF.json
cp $O/pmc_sq$SUF.json $R/profiles/${T}_pmc_sq$SUF.json
if [ "$FULL" = "full" ]; then python bench.py --config $CFG > $O/bench_config$CFG.json 2> $O/bench_config$CFG.err
else python bench.py --config $CFG --cpu-baseline short > $O/bench_config$CFG.json 2> $O/bench_config$CFG.err; fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktg$SUF -o kt -- python $R/bench.py --steps 10 --warmup 3 $Q --no-kernel-profile > $O/bench_profiled_graph_only$SUF.json 2> $O/ktg$SUF.err
cd $R
cp $(find $O/ktg$SUF -name "*kernel_stats.csv" | head -1) $O/kernel_stats_graph_only$SUF.csv
python - <<PY > $O/graph_gaps$SUF.txt
import csv,glob
f=glob.glob("$O/ktg$SUF/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "augment_kernel" in r["Kernel_Name"]]
idx=idx[-10:]
seg=rows[idx[0]:]
wall=(int(rows[idx[-1]]["Start_Timestamp"])-int(rows[idx[0]]["Start_Timestamp"]))/(len(idx)-1)
inner=rows[idx[0]:idx[-1]]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in inner)/(len(idx)-1)
print("config $CFG, last %d steps: kernels/step %.1f  wall/step %.1f us  sum of kernel durations/step %.1f us  gaps/step %.1f us"%(len(idx),len(inner)/(len(idx)-1),wall/1e3,busy/1e3,(wall-busy)/1e3))
fam={}
for r in inner:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    fam[k]=fam.get(k,[0,0]); fam[k][0]+=1; fam[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
n=len(idx)-1
for k,v in sorted(fam.items(),key=lambda kv:-kv[1][1]): print("%-78s %6.1f launches/step %9.1f us/step %7.1f us avg"%(k[:78],v[0]/n,v[1]/1e3/n,v[1]/1e3/v[0]))
PY
rm -rf $O/pmc_fetch$SUF $O/pmc_write$SUF $O/pmc_sq$SUF $O/ktg$SUF
python scratch/r06/gemm_floor_table.py $CFG > $O/floor_config$CFG.txt 2>&1
if [ "$CFG" = "1" ]; then python scratch/r04/family_table.py $O/graph_gaps.txt > $O/family_table.txt 2>&1; fi
