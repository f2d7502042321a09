#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/suite.txt 2>&1; tail -3 $OUT/suite.txt | cut -c1-300
python tools/single_frame_times.py 2>/dev/null | tail -10 > $OUT/single.md; cat $OUT/single.md
tools/qb.sh "c3b d1 c1"
python tools/common_shapes_times.py 2>/dev/null > $OUT/common.md; cat $OUT/common.md
