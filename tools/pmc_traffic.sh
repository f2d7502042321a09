#!/bin/bash
# HBM traffic per launch of each workload's dominant kernel: separate --pmc passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: FETCH_SIZE is doubled on gfx950, both are KiB).  usage: tools/pmc_traffic.sh <tag> "<workloads>"
TAG=$1; WLS=${2:-"c2a c2b c4 c3a c3b c5 c1 d1 d2"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
STEPS=5; WARM=1     # sws_scale_frames() calls per run = STEPS + WARM: the divisor of the per-call sums below
ROOT=$PWD
cd /tmp
for w in $WLS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/traffic_${w}_$c -o res -- python $ROOT/bench.py --workload $w --variants none --no-cpu --steps $STEPS --warmup $WARM > $OUT/traffic_${w}_$c.log 2>&1
  done
done
cd $ROOT
python - "$OUT" "$WLS" $((STEPS + WARM)) <<'PY'
import csv, glob, json, sys, collections
out, wls, ncalls = sys.argv[1], sys.argv[2].split(), int(sys.argv[3])
res = {}
for w in wls:
    # every dispatch of a library kernel (sws_k*) of the run, keyed by the FULL kernel name -- template arguments kept: the luma and the
    # chroma instantiation of one strip kernel are two launches of one call and their bytes ADD (round 5 keyed by the bare name and
    # averaged them: half the traffic for every two-launch workload).  Per call = sum over all dispatches / calls of the run.
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/traffic_{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void ", "").replace("swsk::", "")
                k = k[:k.rindex("(")] if k.endswith(")") and "(" in k else k
                if not k.startswith("sws_k"): continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tot_f = sum(sum(v["FETCH_SIZE"]) for v in acc.values()) / ncalls
    tot_w = sum(sum(v["WRITE_SIZE"]) for v in acc.values()) / ncalls
    per_kernel = {k: {"dispatches_per_call": len(v["FETCH_SIZE"]) / ncalls,
                      "FETCH_SIZE_KiB_raw_per_call": sum(v["FETCH_SIZE"]) / ncalls, "WRITE_SIZE_KiB_per_call": sum(v["WRITE_SIZE"]) / ncalls}
                  for k, v in sorted(acc.items())}
    res[w] = {"kernels": per_kernel, "FETCH_SIZE_KiB_raw": tot_f, "WRITE_SIZE_KiB": tot_w,
              "hbm_bytes_per_launch": int((2 * tot_f + tot_w) * 1024),
              "note": f"separate --pmc passes (FETCH_SIZE, WRITE_SIZE); all sws_k* dispatches of the run summed, / {ncalls} sws_scale_frames() calls "
                      "(= one launch set per call); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md"}
json.dump(res, open(f"{out}/pmc_latest.json", "w"), indent=1)
print(json.dumps({w: {k: v for k, v in r.items() if k != "kernels"} for w, r in res.items()}, indent=1))
PY
