#!/bin/bash
# usage: tools/pmc_run.sh <tag> <workload> "<counters, space separated>"   -> gpurun_out/<tag>/pmc_<workload>.csv (summed per kernel)
TAG=$1; W=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${W}_$i -o res -- python $ROOT/bench.py --workload $W --variants none --no-cpu --steps 5 --warmup 1 $BENCH_ARGS > $OUT/pmc_${W}_$i.log 2>&1
done
cd $ROOT
python - "$OUT" "$W" <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_{w}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "swsk" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} per-dispatch avg {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
