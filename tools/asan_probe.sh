#!/bin/bash
mkdir -p gpurun_out/r06_asan_probe
export ASAN_LOG=$PWD/gpurun_out/r06_asan_probe/report
( timeout 240 tools/asan_env.sh python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -o faulthandler_timeout=100 -k "config" 2>&1 | tail -60 ) > gpurun_out/r06_asan_probe/log.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_asan_probe/log.txt
tail -70 gpurun_out/r06_asan_probe/log.txt
ls gpurun_out/r06_asan_probe
