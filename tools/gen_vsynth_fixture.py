#!/usr/bin/env python3
"""Generate tests/golden/vsynth1_f0_352x288.yuv420p.bin.

Runs the reference's OWN test-video generator (tests/videogen.c, compiled from its own
sources by oracle/ref_build.mk into oracle/_ref/videogen) and keeps frame 0 of the
352x288 yuv420p stream -- the input of fate-sws-yuv-range / fate-sws-yuv-colorspace
(tests/fate/libswscale.mak) and of fate-filter-pixfmts-* (tests/fate-run.sh pixfmts()).
Build container only (needs /root/reference)."""
import os, subprocess, sys, tempfile, zlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "-f", "ref_build.mk"])
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "vsynth1.yuv")
    subprocess.check_call([os.path.join(root, "oracle", "_ref", "videogen"), out])
    data = open(out, "rb").read()
fsz = 352 * 288 * 3 // 2
assert len(data) % fsz == 0
dst = os.path.join(root, "tests", "golden", "vsynth1_f0_352x288.yuv420p.bin.z")
open(dst, "wb").write(zlib.compress(data[:fsz], 9))
print("wrote", dst, len(data[:fsz]), "bytes raw")
# frames 1..3: together with frame 0 they are the first two 352x288 yuv444p pictures that fate-filter-scalechroma reads
# out of the same byte stream (tests/fate/filter-video.mak:533-535)
import lzma
dst2 = os.path.join(root, "tests", "golden", "vsynth1_f1-3_352x288.yuv420p.bin.xz")
open(dst2, "wb").write(lzma.compress(data[fsz:4 * fsz], preset=9 | lzma.PRESET_EXTREME))
print("wrote", dst2, os.path.getsize(dst2), "bytes")
# frame 4: with frames 0..3 the five pictures FATE's video_filter() recipes encode (-frames:v 5: filter-null / -scale200 / -scale500 / -crop_scale ...)
dst3 = os.path.join(root, "tests", "golden", "vsynth1_f4_352x288.yuv420p.bin.xz")
open(dst3, "wb").write(lzma.compress(data[4 * fsz:5 * fsz], preset=9 | lzma.PRESET_EXTREME))
print("wrote", dst3, os.path.getsize(dst3), "bytes")
