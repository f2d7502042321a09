#!/bin/bash
# per-kernel durations of one bench workload: tools/kstats.sh <workload> [bench args] -> markdown summary on stdout
W=$1; shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$PWD/gpurun_out/kstats_$W; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o res -- python $ROOT/bench.py --workload $W --variants none --no-cpu --steps 20 --warmup 3 "$@" > $OUT/log 2>&1)
db=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" "$W: rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps 20 --warmup 3 $*"
