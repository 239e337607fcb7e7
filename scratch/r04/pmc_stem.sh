cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_stem}; mkdir -p $O
python $R/scratch/r04/stem_bench.py 2>&1 | grep -v amdgpu
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python $R/scratch/r04/stem_bench.py > $O/$tag.log 2>&1 )
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "stem" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({c[3:]: round(sum(v)/len(v)/16384, 1) for c,v in sorted(agg.items())}, "(per wave-tile; 16384 wave-tiles)")
PY
