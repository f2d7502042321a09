#!/bin/bash
# Round 6, first GPU call after the closure: (1) the restructured host side under the GPU suite, (2) the experiment instantiations against the oracle, (3) A/B bench lines.
# usage (GPU box): tools/r06_experiments.sh   -> gpurun_out/r06x/
OUT=$PWD/gpurun_out/r06x; mkdir -p $OUT
{
echo "== GPU suite, -n 4"; date
timeout 900 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider 2>&1 | tail -6
echo "== experiment parity"; date
timeout 600 python tools/exp_parity.py 2>&1 | tail -40
echo "== bench A/B"; date
B="python bench.py --variants none --no-cpu --steps 20 --warmup 3"
for o in "" "--opt exp1=1" "--opt exp0=3 --opt exp1=1 --opt exp2=5" "--opt strip_cols_l=2 --opt strip_cols_c=1 --opt exp2=7 --opt strip_waves=7168" "--opt strip_cols_l=2 --opt strip_cols_c=1 --opt exp2=7 --opt exp1=1 --opt strip_waves=7168" "--opt strip_cols_l=3 --opt exp4=1 --opt strip_waves=5120" "--opt strip_cols_l=3 --opt exp4=1 --opt exp1=1 --opt strip_waves=5120"; do
  for b in 64 8; do echo "c3b x$b [$o]"; timeout 300 $B --workload c3b --batch $b $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms_avg'], d['config']['path'])"; done
done
# (C1's planner picks 5 columns per lane for both plane classes: 76 VGPRs luma, 114 chroma -- narrower chroma strips buy waves per SIMD)
for o in "" "--opt exp3=1" "--opt strip_cols_auto=0 --opt strip_cols_l=5 --opt strip_cols_c=3" "--opt strip_cols_auto=0 --opt strip_cols_l=5 --opt strip_cols_c=2" "--opt strip_cols_auto=0 --opt strip_cols_l=3 --opt strip_cols_c=3"; do
  for b in 256 64; do echo "c1 x$b [$o]"; timeout 300 $B --workload c1 --batch $b $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms_avg'], d['config']['path'])"; done
done
echo "== default bench line"; date
timeout 900 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
date
} > $OUT/log.txt 2>&1
tail -80 $OUT/log.txt
