#!/bin/bash
# Builds the profiling variant of the library (-DSWS_HIP_PROFILING: stage switches with WRONG results, "debug" option) next to the
# product build: librempeg_amd/lib/prof/libswscale_hip.so.  Use with SWS_HIP_LIBRARY=$PWD/librempeg_amd/lib/prof/libswscale_hip.so
cd "$(dirname "$0")/../librempeg_amd/csrc" && make -s -j8 EXTRA=-DSWS_HIP_PROFILING OUT=../lib/prof/libswscale_hip.so OBJ=../lib/prof/obj
