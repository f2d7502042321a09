#!/bin/bash
# HBM traffic per launch of each workload's dominant kernel: separate --pmc passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: FETCH_SIZE is doubled on gfx950, both are KiB).  usage: tools/pmc_traffic.sh <tag> "<workloads>"
TAG=$1; WLS=${2:-"c2a c2b c4 c3a c3b c5 c1 d1 d2"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for w in $WLS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/traffic_${w}_$c -o res -- python $ROOT/bench.py --workload $w --variants none --no-cpu --steps 5 --warmup 1 > $OUT/traffic_${w}_$c.log 2>&1
  done
done
cd $ROOT
python - "$OUT" "$WLS" <<'PY'
import csv, glob, json, sys, collections, subprocess
out, wls = sys.argv[1], sys.argv[2].split()
res = {}
for w in wls:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/traffic_{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("swsk::", "")
                if not k.startswith("sws_k"): continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the workload's launches: sum over its kernels (luma + chroma launches count as one unit of work)
    line = json.loads(open(f"{out}/traffic_{w}_FETCH_SIZE.log").read().strip().splitlines()[-1]) if False else None
    tot_f = sum(sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) for v in acc.values() if v.get("FETCH_SIZE"))
    tot_w = sum(sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) for v in acc.values() if v.get("WRITE_SIZE"))
    res[w] = {"kernels": sorted(acc.keys()), "FETCH_SIZE_KiB_raw": tot_f, "WRITE_SIZE_KiB": tot_w,
              "hbm_bytes_per_launch": int((2 * tot_f + tot_w) * 1024),
              "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), per-dispatch averages summed over the workload's kernels; "
                      "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md"}
json.dump(res, open(f"{out}/pmc_latest.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
