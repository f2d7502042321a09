#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/suite.txt 2>&1; tail -8 $OUT/suite.txt | cut -c1-300
timeout 300 python tools/common_shapes_times.py > $OUT/common.md 2>$OUT/common.err; cat $OUT/common.md
