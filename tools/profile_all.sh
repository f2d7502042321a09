#!/bin/bash
# Run every BASELINE workload of bench.py once under rocprofv3 --kernel-trace --stats and once plain;
# results (JSON lines + rocpd summaries) go to gpurun_out/<tag>/.  usage: tools/profile_all.sh <tag>
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for w in ${SWS_PROFILE_WLS:-c2a c2b c4 c3a c3b c5 c1 d1 d2 e1 e2 r1 r2 w1 f1 u1}; do
    python bench.py --workload $w --variants none --no-cpu --steps 20 --warmup 3 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
    (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_$w" -o res -- python "$OLDPWD/bench.py" --workload $w --variants none --no-cpu --steps 20 --warmup 3 > "$OUT/prof_$w.log" 2>&1)
    db=$(find "$OUT/prof_$w" -name "*.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$TAG $w: rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 20 --warmup 3" > "$OUT/kernel_stats_$w.md"
    rm -rf "$OUT/prof_$w"
done
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cat "$OUT"/bench_*.json
