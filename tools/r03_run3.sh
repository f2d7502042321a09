#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rgbread.py tests/test_gpu_cascade_slices.py tests/test_gpu_rgbsrc.py -q > $OUT/suite.txt 2>&1; tail -8 $OUT/suite.txt | cut -c1-300
timeout 300 python tools/common_conversions_times.py > $OUT/conv.txt 2>$OUT/conv.err; cat $OUT/conv.txt
timeout 300 python tools/aux_kernel_times.py > $OUT/aux.txt 2>$OUT/aux.err; cat $OUT/aux.txt
timeout 300 python tools/common_shapes_times.py > $OUT/common.md 2>$OUT/common.err; cat $OUT/common.md
