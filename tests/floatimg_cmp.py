"""Restatement of the reference's fate-sws-floatimg-cmp (libswscale/tests/floatimg_cmp.c; known answers in
tests/golden/fate_sws_floatimg_cmp.txt = the reference's tests/ref/fate/sws-floatimg-cmp):

a 96x96 gbrpf32le picture of av_lfg floats (av_lfg_init(&rand, 1), libavutil/lfg.c:29-43, lfg.h:53-57) goes to each of 34
formats and back with sws_getContext(..., SWS_BILINEAR, NULL, NULL, NULL); the test prints the average / minimum / maximum
absolute difference of the round trip with "%f".  Pins planar_rgbf32_to_y / _to_uv (input.c:1300-1334) and the float planar
RGB writer (output.c:2533-2605) against reference answers.  Used with the oracle (CPU) and with the HIP library (GPU)."""
import hashlib
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fate_sws_floatimg_cmp.txt")
W = H = 96
SWS_BILINEAR = 2


def av_lfg_floats(n, seed=1):
    """n values of (float)av_lfg_get(&rand) / (float)UINT32_MAX after av_lfg_init(&rand, seed)."""
    state = [0] * 64
    tmp = bytearray(16)
    for i in range(8, 64, 4):
        tmp[0:4] = struct.pack("<I", seed)
        tmp[4] = i
        tmp = bytearray(hashlib.md5(bytes(tmp)).digest())
        state[i:i + 4] = struct.unpack("<4I", bytes(tmp))
    out = np.empty(n, dtype=np.float32)
    idx = 0
    for k in range(n):
        a = (state[(idx - 24) & 63] + state[(idx - 55) & 63]) & 0xFFFFFFFF
        state[idx & 63] = a
        idx += 1
        out[k] = np.float32(a) / np.float32(4294967295.0)
    return out


def golden():
    """[(format, avg, min, max)] in the reference's order."""
    lines = open(GOLDEN).read().split("\n")
    out = []
    for i in range(0, len(lines) - 3, 4):
        fmt = lines[i].split(" -> ")[1]
        out.append((fmt, lines[i + 1].split(": ")[1], lines[i + 2].split(": ")[1], lines[i + 3].split(": ")[1]))
    return out


def source_planes():
    v = av_lfg_floats(3 * W * H).reshape(3, H, W)   # planes 0, 1, 2 of gbrpf32le, filled one after the other
    return [v[0], v[1], v[2]]


def stats(src_planes, out_planes):
    """the test's statistics: sum of float32 differences in double, min / max in float32, printed with %f"""
    total, count, mn, mx = 0.0, 0, np.float32(3.4028235e38), np.float32(-3.4028235e38)
    for a, b in zip(src_planes, out_planes):
        d = np.abs(a.astype(np.float32) - b.astype(np.float32)).astype(np.float32)
        total += float(d.astype(np.float64).sum())
        count += d.size
        mn, mx = min(mn, d.min()), max(mx, d.max())
    return "%f" % (total / count), "%f" % mn, "%f" % mx
