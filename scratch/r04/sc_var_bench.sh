#!/bin/bash
# scratch/sc_bench.py for every what-if library named (built by scratch/r04/sc_variants.sh); prints a side-by-side table
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-sc_var}; shift; mkdir -p $OUT
for t in "$@"; do EXP=$t timeout 300 python scratch/sc_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_$t.txt; done
python - "$OUT" "$@" <<'PY'
import sys,re
out=sys.argv[1]; tags=sys.argv[2:]
rows={}
for t in tags:
    for l in open("%s/bench_%s.txt"%(out,t)):
        m=re.match(r"EXP=\S+ (.*?)\s+([0-9.]+) us",l)
        if m: rows.setdefault(m.group(1).strip(),{})[t]=float(m.group(2))
print("%-34s"%"layer"+"".join("%9s"%t for t in tags))
for k,v in rows.items(): print("%-34s"%k+"".join("%9.1f"%v.get(t,0) for t in tags))
PY
