#!/bin/bash
# A/B of sws_k_rgbsrc_unity2's banding on the capture -> encoder shapes (options through SWSOPT_*): tools/exp_rgbsrc2.sh
run() { echo "== $*"; env "$@" python tools/common_shapes_times.py 2>&1 | grep -E "rgbsrc" | cut -d'|' -f2,4,6; }
run SWSOPT_NO_RGBSRC2=1
run X=1
run SWSOPT_STRIP_MIN_ROWS=2
run SWSOPT_STRIP_MIN_ROWS=8
run SWSOPT_STRIP_MIN_ROWS=16
run SWSOPT_STRIP_WAVES=2048
run SWSOPT_STRIP_WAVES=8192
run SWSOPT_STRIP_COLS_AUTO=0 SWSOPT_STRIP_COLS_L=1
run SWSOPT_STRIP_COLS_AUTO=0 SWSOPT_STRIP_COLS_L=1 SWSOPT_STRIP_MIN_ROWS=8
run SWSOPT_STRIP_COLS_AUTO=0 SWSOPT_STRIP_COLS_L=2 SWSOPT_STRIP_MIN_ROWS=8
