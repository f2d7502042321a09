#!/bin/bash
# Runs a command (normally pytest) with the sanitizer builds of the product's host side and of the oracle:
#   make -C librempeg_amd/csrc asan ; make -C oracle asan ; tools/asan_env.sh python -m pytest tests -m gpu -x -q
# Both are built with clang and -shared-libsan, so ONE AddressSanitizer runtime (preloaded into python) watches numpy's buffers,
# the ctypes call frames, the library's host code and the oracle.  UBSan reports are logged and the run goes on (tools/asan_summary.py lists the unique sites).
#
# SWS_ASAN_RUNTIME=gnu (the choice for a GPU box): the same instrumented objects under gcc's libasan + libubsan (make -C librempeg_amd/csrc asan_gnu ; make -C oracle asan_gnu).
# ROCm's clang runtime interposes hsa_amd_memory_pool_allocate & co. for device ASan and fails every device allocation of the uninstrumented ROCr / torch HIP of this
# image; gcc's runtime knows nothing of HSA.  (One clang-only UBSan handler is stubbed: tools/bin/libubsan_fn_stub.so.)
# PYTHONMALLOC=malloc: python's small-object arenas would hide ctypes buffers from the sanitizer (tools/asan_selftest.py checks that an overflow IS seen).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONMALLOC=malloc
# SWS_HIPSTUB=1 (a box WITHOUT a GPU): the HIP runtime's test double (tests/hipstub: bounds-checked host memory as device memory, launches dropped) goes behind the
# sanitizer runtime in the preload list, so that the product's device-state / upload / ring / staging / teardown code runs under ASan here (tools/hipstub_hunt.py)
if [ "${SWS_HIPSTUB:-0}" = 1 ]; then
  make -s -C $ROOT/tests/hipstub || exit 1
  LD_PRELOAD=$ROOT/tests/hipstub/libhipstub.so${LD_PRELOAD:+:$LD_PRELOAD}
fi
if [ "${SWS_ASAN_RUNTIME:-clang}" = gnu ]; then
  STUB=$ROOT/tools/bin/libubsan_fn_stub.so
  [ -f $STUB ] || { mkdir -p $ROOT/tools/bin; printf '#include <stdio.h>\nvoid __ubsan_handle_function_type_mismatch(void *d, void *v) { (void)d; (void)v; fprintf(stderr, "ubsan: function type mismatch\\n"); }\nvoid __ubsan_handle_function_type_mismatch_abort(void *d, void *v) { __ubsan_handle_function_type_mismatch(d, v); }\n' > /tmp/ubsan_fn_stub.c; gcc -O2 -shared -fPIC -o $STUB /tmp/ubsan_fn_stub.c; }
  export SWS_HIP_LIBRARY=$ROOT/librempeg_amd/lib_asan_gnu/libswscale_hip.so
  export SWS_ORACLE_LIBRARY=$ROOT/oracle/asan_gnu/libsws_oracle.so
  export ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:log_path=${ASAN_LOG:-/tmp/asan_report}:detect_odr_violation=0}
  export UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1:halt_on_error=0:log_path=${ASAN_LOG:-/tmp/asan_report}}
  export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so):$STUB${LD_PRELOAD:+:$LD_PRELOAD}
  exec "$@"
fi
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export SWS_HIP_LIBRARY=$ROOT/librempeg_amd/lib_asan/libswscale_hip.so
export SWS_ORACLE_LIBRARY=$ROOT/oracle/asan/libsws_oracle.so
# python "leaks" by design; the HSA runtime maps the shadow gap; abort on the first report so that pytest shows the test
export ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1:log_path=${ASAN_LOG:-/tmp/asan_report}:print_stacktrace=1:detect_odr_violation=0}
export UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1:halt_on_error=0:log_path=${ASAN_LOG:-/tmp/asan_report}}
# ROCm's ASan runtime interposes the HSA allocation calls for device ASan and fails them on this image's uninstrumented runtime:
# tools/asan_hsa_passthrough.c hands them straight to libhsa-runtime64 (built here if missing; must come FIRST in the preload list)
SHIM=$ROOT/tools/bin/libasan_hsa_passthrough.so
[ -f $SHIM ] || { mkdir -p $ROOT/tools/bin; gcc -O2 -shared -fPIC -I/opt/rocm/include -o $SHIM $ROOT/tools/asan_hsa_passthrough.c -ldl; }
export LD_PRELOAD=$SHIM:$RT${LD_PRELOAD:+:$LD_PRELOAD}
exec "$@"
