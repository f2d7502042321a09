#!/bin/bash
# The sanitizer pass of round 6 on a GPU box: the full GPU suite, then the random generators, under
# the sanitizer builds (tools/asan_env.sh; on a GPU box with the gcc runtime: SWS_ASAN_RUNTIME=gnu, set below), four pytest workers on the one GPU -- the condition
# of every rare event of DESIGN.md section 8.  Run tools/asan_probe.sh first (bounded: does one test file get through?).
# usage: tools/asan_gpu_run.sh <tag> [N per generator] [seed]
TAG=${1:-asan}; N=${2:-8000}; SEED=${3:-606}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export ASAN_LOG=$OUT/report
export SWS_ASAN_RUNTIME=${SWS_ASAN_RUNTIME:-gnu}
{
  echo "== suite under ASan+UBSan, -n 4"; date
  tools/asan_env.sh python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider 2>&1 | tail -15
  echo "== random generators N=$N seed=$SEED under ASan+UBSan, -n 4"; date
  SWS_RANDOM_N=$N SWS_RANDOM_SEED=$SEED tools/asan_env.sh python -m pytest tests/test_gpu_random.py tests/test_gpu_guard_bands.py -q -n 4 -p no:cacheprovider 2>&1 | tail -15
  date
  echo "== sanitizer reports"
  python tools/asan_summary.py $OUT/report
} > $OUT/log.txt 2>&1
tail -40 $OUT/log.txt
