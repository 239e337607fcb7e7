# instruction counts of the small-channel kernels under the what-if builds (EXP list in $2..)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_insts}; shift
mkdir -p $O
for e in "$@"; do
  ( cd /tmp && EXP=$e rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/e$e -o p -- python $R/scratch/sc_bench.py > $O/e$e.log 2>&1 )
done
python - <<PY
import csv, glob, collections
O="$O"
for f in sorted(glob.glob(O+"/e*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "conv_sc" not in k: continue
        agg[(k,r.get("Grid_Size"),r.get("LDS_Block_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==",f.split("/")[-2])
    for (k,g,l),d in sorted(agg.items()):
        print(k[28:],g,l,len(list(d.values())[0]), {c[3:]:round(sum(v)/len(v)/1e3) for c,v in d.items()})
PY
