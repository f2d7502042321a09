#!/bin/bash
# A/B of the short strip instantiations on C1 (and neighbours): tools/exp_short.sh
python -m pytest tests/test_gpu_strip_short.py -x -q -m gpu 2>&1 | tail -8
for b in 256 64; do
for o in "" "--opt no_strip_dma8=1" "--opt no_strip_short=1" "--opt strip_cols_auto=0 --opt strip_cols_l=5 --opt strip_cols_c=3" "--opt strip_cols_auto=0 --opt strip_cols_l=5 --opt strip_cols_c=5" \
         "--opt strip_cols_auto=0 --opt strip_cols_l=7 --opt strip_cols_c=5" "--opt strip_cols_auto=0 --opt strip_cols_l=4 --opt strip_cols_c=2" "--opt strip_dma8_depth=16" "--opt strip_dma8_depth=40"; do
  echo "== c1 x$b $o"; tools/qb.sh c1 --batch $b $o
done; done
