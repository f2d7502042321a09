#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/suite.txt 2>&1; tail -6 $OUT/suite.txt | cut -c1-300
python tools/single_frame_times.py 2>/dev/null | tail -10
echo "unfused:"; SWSOPT_DEBUG=65536 python tools/single_frame_times.py 2>/dev/null | grep -E "strip|rgbread"
for b in 8 32; do echo c3b batch $b fused; tools/qb.sh c3b --batch $b; echo c3b batch $b unfused; tools/qb.sh c3b --batch $b --opt debug=65536; done
echo c1 strip fused; tools/qb.sh c1 --opt strip_min_w=0; echo c1 tile; tools/qb.sh c1
python tools/common_shapes_times.py 2>/dev/null > $OUT/common.md; cat $OUT/common.md
