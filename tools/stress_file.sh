#!/bin/bash
# runs one test file in a loop in P processes sharing the GPU for S seconds: the condition of every rare parity event.  usage: tools/stress_file.sh <pytest args> -- P S
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
P=${1:-4}; S=${2:-200}
mkdir -p gpurun_out/stress
END=$(( $(date +%s) + S ))
for k in $(seq $P); do
  ( n=0; f=0; while [ $(date +%s) -lt $END ]; do n=$((n+1)); python -m pytest "${ARGS[@]}" -q -x -p no:cacheprovider > gpurun_out/stress/p$k.log 2>&1 || { f=$((f+1)); cp gpurun_out/stress/p$k.log gpurun_out/stress/fail_p${k}_$n.log; }; done; echo "process $k: $n runs, $f failed" ) &
done
wait
