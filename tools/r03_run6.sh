#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/suite.txt 2>&1; tail -12 $OUT/suite.txt | cut -c1-300
timeout 300 python tools/common_conversions_times.py > $OUT/conv.txt 2>$OUT/conv.err; tail -3 $OUT/conv.txt
