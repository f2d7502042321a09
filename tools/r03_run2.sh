#!/bin/bash
# second GPU call of round 3: suite on the new paths (rgbread, dma8, cascade slices, packed layout arithmetic), tables, experiments
set -u
OUT=$PWD/gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/suite.txt 2>&1; tail -15 $OUT/suite.txt | cut -c1-300
timeout 300 python tools/layout_times.py > $OUT/layout.md 2>$OUT/layout.err; cat $OUT/layout.md
timeout 300 python tools/common_shapes_times.py > $OUT/common.md 2>$OUT/common.err; cat $OUT/common.md
timeout 600 tools/qb.sh "c2a c2b c4 c3a c3b c5 c1 d1 d2" > $OUT/qb.txt 2>&1; cat $OUT/qb.txt
{
echo "c3b two streams"; tools/qb.sh c3b --opt debug=256
echo "c3b cols 2/1 wpe4"; tools/qb.sh c3b --opt strip_cols_l=2 --opt strip_cols_c=1
echo "c3b cols 2/1 wpe6"; tools/qb.sh c3b --opt strip_cols_l=2 --opt strip_cols_c=1 --opt debug=24576 --opt strip_waves=6144
echo "c3b cols 2/1 wpe8"; tools/qb.sh c3b --opt strip_cols_l=2 --opt strip_cols_c=1 --opt debug=32768 --opt strip_waves=8192
echo "c3b cols 2/1 wpe8 two streams"; tools/qb.sh c3b --opt strip_cols_l=2 --opt strip_cols_c=1 --opt debug=33024 --opt strip_waves=8192
echo "c3b cols 4/2 waves 8192"; tools/qb.sh c3b --opt strip_waves=8192
echo "c3b no dma"; tools/qb.sh c3b --opt no_strip_dma=1
for cl in 4 2; do for cc in 2 1; do for sw in 2048 4096 8192; do echo "c1 strip cols $cl/$cc strip_waves=$sw"; tools/qb.sh c1 --opt strip_min_w=0 --opt strip_waves=$sw --opt strip_cols_l=$cl --opt strip_cols_c=$cc; done; done; done
echo "c1 strip no dma"; tools/qb.sh c1 --opt strip_min_w=0 --opt no_strip_dma=1
} > $OUT/exp.txt 2>&1; cat $OUT/exp.txt
