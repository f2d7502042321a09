#!/bin/bash
# bounded probe of the sanitizer set-up on a GPU box: the self-test (is an overflow seen?), then ONE test file with python tracebacks dumped after 100 s and a 240 s limit.
export SWS_ASAN_RUNTIME=${SWS_ASAN_RUNTIME:-gnu}
mkdir -p gpurun_out/r06_asan_probe
export ASAN_LOG=$PWD/gpurun_out/r06_asan_probe/report
( timeout 60 tools/asan_env.sh python tools/asan_selftest.py 2>&1 | tail -3; timeout 240 tools/asan_env.sh python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -o faulthandler_timeout=100 -k "config" 2>&1 | tail -60 ) > gpurun_out/r06_asan_probe/log.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_asan_probe/log.txt
tail -70 gpurun_out/r06_asan_probe/log.txt
ls gpurun_out/r06_asan_probe
