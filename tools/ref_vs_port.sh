#!/bin/bash
# port / reference, one thread, THIS container: builds a C-only libswscale of /root/reference under /tmp (SURVEY Appendix B recipe 1:
# --disable-asm; the build and its objects stay in /tmp, nothing of them enters the repo), times the BASELINE configurations with the real
# reference (tools/ref/ref_time.c, best of N) and with oracle/ (the port bench.py's cpu_baseline runs), and writes the two columns + their
# ratio to profiles/ref_vs_port.json -- a data file bench.py reads on the GPU box to scale the port's rate into a reference-equivalent one.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${SWS_REFERENCE_ROOT:-/root/reference}
B=${REFBUILD:-/tmp/refbuild}
if [ ! -f $B/libswscale/libswscale.a ]; then
  mkdir -p $B && cd $B
  $REF/configure --disable-everything --disable-programs --disable-doc --disable-avcodec --disable-avformat --disable-avdevice \
      --disable-swresample --enable-swscale --disable-asm --disable-autodetect --enable-static --disable-shared > configure.log 2>&1
  make -j8 libswscale/libswscale.a libavutil/libavutil.a > make.log 2>&1
fi
gcc -O2 -I$REF -I$B -o $B/ref_time $ROOT/tools/ref/ref_time.c $B/libswscale/libswscale.a $B/libavutil/libavutil.a -lm -lpthread
make -s -C $ROOT/oracle
cd $ROOT && python3 tools/ref/ref_vs_port.py $B/ref_time
