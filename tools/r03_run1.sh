#!/bin/bash
# first GPU call of round 3: suite, layout / common-shape tables, every bench workload, strip kernel on C1
set -u
OUT=$PWD/gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/suite.txt 2>&1; tail -3 $OUT/suite.txt
timeout 300 python tools/layout_times.py > $OUT/layout.md 2>$OUT/layout.err; cat $OUT/layout.md
timeout 300 python tools/common_shapes_times.py > $OUT/common.md 2>$OUT/common.err; cat $OUT/common.md
timeout 600 tools/qb.sh "c2a c2b c4 c3a c3b c5 c1 d1 d2" > $OUT/qb.txt 2>&1; cat $OUT/qb.txt
for cl in 4 2; do for cc in 2 1; do for sw in 4096 8192 16384; do echo "c1 strip cols $cl/$cc strip_waves=$sw"; timeout 120 tools/qb.sh c1 --opt strip_min_w=0 --opt strip_waves=$sw --opt strip_cols_l=$cl --opt strip_cols_c=$cc; done; done; done > $OUT/c1_strip.txt 2>&1; cat $OUT/c1_strip.txt
