# SQ counters of the small-channel kernels (scratch/scw_bench.py, scratch/sc_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_scw}
mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f2)
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o p -- python $R/scratch/${2:-scw_bench.py} > $O/pmc_$tag.log 2>&1 )
done
python - <<PY
import csv, glob, collections
O="$O"
for f in sorted(glob.glob(O+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "conv_sc" not in k: continue
        agg[(k,r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==",f.split("/")[-3])
    for (k,g),d in agg.items():
        print(k,g,len(list(d.values())[0]), {c:round(sum(v)/len(v)) for c,v in d.items()})
PY
