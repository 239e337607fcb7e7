# per-dispatch L2-miss traffic of the weight-gradient launches (FETCH_SIZE, gfx950: x 2): bash scratch/r04/pmc_wgrad_traffic.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile --sustain 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
python - <<PY > $O/wgrad_fetch_per_dispatch.txt
import csv, glob
p = glob.glob("$O/pmc_fetch/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == "FETCH_SIZE" and "wgrad" in r["Kernel_Name"] and ("group" in r["Kernel_Name"])]
for r in rows[-24:]:
    print("%-60s grid %6s  fetch %8.1f MB" % (r["Kernel_Name"][:60], r.get("Grid_Size", "?"), 2 * float(r["Counter_Value"]) / 1024))
PY
rm -rf $O/pmc_fetch
