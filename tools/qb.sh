#!/bin/bash
# quick bench: tools/qb.sh "<workloads>" [extra bench.py args] -> one brief line per workload
WLS=${1:-"c2a c2b c4 c3b"}; shift
for w in $WLS; do
  python bench.py --workload $w --variants none --no-cpu --steps 40 --warmup 5 "$@" 2>/dev/null | python tools/bench_brief.py $w
done
