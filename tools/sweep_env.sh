#!/bin/bash
# usage: tools/sweep_env.sh <ENVVAR> "<values>" "<workloads>"  -> one line per (workload, value): ms/step, kernel ms, roofline fraction
VAR=$1; VALS=$2; WLS=$3
for w in $WLS; do
  for v in $VALS; do
    env $VAR=$v python bench.py --workload $w --variants none --no-cpu --steps 30 --warmup 3 2>/dev/null | python tools/bench_brief.py "$w $VAR=$v"
  done
done
