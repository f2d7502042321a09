#!/bin/bash
# usage: tools/sweep2.sh "<workloads>" "ENV1=a ENV2=b" "ENV1=c" ...   one bench line per (workload, env set)
WLS=$1; shift
for w in $WLS; do
  for e in "$@"; do
    env $e python bench.py --workload $w --variants none --no-cpu --steps 30 --warmup 3 2>/dev/null | python tools/bench_brief.py "$w [$e]"
  done
done
