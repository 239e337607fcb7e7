# rocprofv3 kernel stats of the step graphs of BASELINE configs[3] / [4] (and Linknet): per-kernel time of 25 graph steps each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_other; mkdir -p $O
for c in FPN:bf16 PSPNet:bf16 Linknet:bf16; do
  n=$(echo $c | cut -d: -f1 | tr 'A-Z' 'a-z')
  cd /tmp
  PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o k -- python $R/scratch/other_configs_bench.py $c > $O/$n.json 2> $O/$n.err
  cd $R
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1)
  python - "$f" "$O/${n}_kernel_stats_top.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = open(sys.argv[2], "w")
out.write("kernel (rocprofv3 --kernel-trace --stats over 25 graph steps + set-up), share of kernel time, calls, average us\n")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    out.write("%-90s %5.1f %%  %7d calls  %8.1f us\n" % (r["Name"].split("(")[0].replace("void ", "")[:90], 100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/$n
  cat $O/$n.json | grep workload | cut -c1-120
  head -12 $O/${n}_kernel_stats_top.txt
done
