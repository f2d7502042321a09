#!/bin/bash
# A/B of the one-launch form of scaled packed-RGB sources (sws_k_strip_rgbsrc) against the reader pre-pass + two strip launches, and its banding
# (options through SWSOPT_*): tools/exp_striprgbsrc.sh
run() { echo "== $*"; env "$@" python tools/common_shapes_times.py 2>&1 | grep -E "rgb24 1920x1080 -> yuv420p 1280|bgra 3840x2160 -> yuv420p 1920|rgbsrc|rgbread" | cut -d'|' -f2,3,4,6; }
run SWSOPT_NO_STRIP_RGBSRC=1
run X=1
run SWSOPT_STRIP_MIN_ROWS=2
run SWSOPT_STRIP_MIN_ROWS=8
run SWSOPT_STRIP_MIN_ROWS=16
run SWSOPT_STRIP_WAVES=2048
run SWSOPT_STRIP_WAVES=8192
run SWSOPT_STRIP_SHORT_WAVES=2
run SWSOPT_STRIP_SHORT_WAVES=4
