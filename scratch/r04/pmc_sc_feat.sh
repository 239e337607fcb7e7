# instruction counts per wave-tile of the 16->16 @512 small-channel layer, one feature variant per rocprofv3 run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_sc_feat}; mkdir -p $O
for f in plain pbn stats bnb; do
  ( cd /tmp && ONLY=$f rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/$f -o p -- python $R/scratch/sc_bench.py > $O/$f.log 2>&1 )
done
python - <<PY
import csv, glob, collections
O="$O"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_sc" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2].ljust(8), {c[3:]: round(sum(v)/len(v)/65536, 1) for c,v in sorted(agg.items())}, "(per wave-tile; 65536 wave-tiles)")
PY
