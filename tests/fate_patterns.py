"""Input pictures of the reference's fate-pixfmt tests, restated as data generators (test infrastructure).

* vsynth1 frame 0: committed fixture produced by the reference's tests/videogen.c (tools/gen_vsynth_fixture.py);
* yuvtestsrc / rgbtestsrc at 352x288: three horizontal bands, each a left-to-right ramp c = (1<<depth)*x/w in one
  component (libavfilter/vsrc_testsrc.c:1111-1131 rgbtest_fill_picture, :1290-1312 yuvtest_fill_picture).
"""
import hashlib
import json
import os
import zlib

import numpy as np

import oracle_lib as OL

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 352, 288
SWS_FLAGS = OL.SWS_BICUBIC | OL.SWS_ACCURATE_RND | OL.SWS_BITEXACT   # tests/fate-run.sh:258 + the filter's default scaler
FMT_OF = {"gray": "gray8", "rgb32": "bgra", "rgb48": "rgb48le", "rgb565": "rgb565le", "rgb555": "rgb555le"}                           # lavu pixfmt aliases on little endian
BASE_FMT = {"yuv444p": "yuv444p", "rgb24": "rgb24", "yuv444p10": "yuv444p10le", "yuv444p12": "yuv444p12le",
            "yuv444p16": "yuv444p16le", "nv24": "nv24", "p410": "p410le", "p412": "p412le", "p416": "p416le",
            "gbrp": "gbrp", "gbrp10": "gbrp10le", "gbrp12": "gbrp12le", "gbrp16": "gbrp16le", "rgb48": "rgb48le"}
GOLDEN = json.load(open(os.path.join(HERE, "golden", "fate_pixfmt_md5.json")))


def vsynth_frame():
    raw = zlib.decompress(open(os.path.join(HERE, "golden", "vsynth1_f0_352x288.yuv420p.bin.z"), "rb").read())
    f = OL.Frame("yuv420p", W, H)
    o = 0
    for a, rb in zip(f.planes, f.row_bytes):
        n = rb * a.shape[0]
        a[:, :rb] = np.frombuffer(raw[o:o + n], np.uint8).reshape(a.shape[0], rb)
        o += n
    return f


def vsynth_yuv444_pictures(n=2):
    """fate-filter-scalechroma reads tests/data/vsynth1.yuv (the yuv420p videogen stream) as 352x288 *yuv444p* pictures
    (tests/fate/filter-video.mak:533-535): picture k is bytes [k*304128, (k+1)*304128) = videogen frames 2k and 2k+1."""
    import lzma
    raw = zlib.decompress(open(os.path.join(HERE, "golden", "vsynth1_f0_352x288.yuv420p.bin.z"), "rb").read()) + \
        lzma.decompress(open(os.path.join(HERE, "golden", "vsynth1_f1-3_352x288.yuv420p.bin.xz"), "rb").read())
    out = []
    for k in range(n):
        fr = OL.Frame("yuv444p", W, H)
        d = np.frombuffer(raw[k * W * H * 3:(k + 1) * W * H * 3], np.uint8).reshape(3, H, W)
        for i in range(3):
            fr.planes[i][:, :W] = d[i]
        out.append(fr)
    return out


SCALECHROMA_CRC = [0x77bb80f8, 0x3a21f6e8]   # tests/ref/fate/filter-scalechroma, frames 0 and 1


def band_of(y):
    return 0 if 3 * y < H else (1 if 3 * y < 2 * H else 2)


def yuvtestsrc(depth, semi=False):
    """yuvtest_fill_picture + yuvtest_put_pixel (vsrc_testsrc.c:1201-1312); semi-planar formats (nv24, p41x) keep the
    value in the high bits: v << (16 - depth)."""
    if semi:
        fmt = "nv24" if depth == 8 else f"p4{depth}le"
    else:
        fmt = "yuv444p" if depth == 8 else f"yuv444p{depth}le"
    f = OL.Frame(fmt, W, H)
    ramp = ((1 << depth) * np.arange(W) // W)
    dt = np.uint8 if depth == 8 else np.dtype("<u2")
    planes = [np.full((H, W), 1 << (depth - 1), dt) for _ in range(3)]
    for y in range(H):
        planes[band_of(y)][y, :] = ramp
    if semi:
        if depth > 8:
            planes = [(p.astype(np.uint32) << (16 - depth)).astype(dt) for p in planes]
        uv = np.empty((H, 2 * W), dt)
        uv[:, 0::2], uv[:, 1::2] = planes[1], planes[2]
        planes = [planes[0], uv]
    for a, p in zip(f.planes, planes):
        a[:, :p.shape[1] * p.itemsize] = p.view(np.uint8).reshape(H, -1)
    return f


def rgbtestsrc(planar_depth=0):
    """rgbtest_fill_picture (vsrc_testsrc.c:1111-1131): c = (1 << max(depth, 8)) * x / w in the band's channel."""
    if planar_depth:
        fmt = "gbrp" if planar_depth == 8 else f"gbrp{planar_depth}le"
        f = OL.Frame(fmt, W, H)
        dt = np.uint8 if planar_depth == 8 else np.dtype("<u2")
        ramp = ((1 << planar_depth) * np.arange(W) // W).astype(dt)
        rgb = [np.zeros((H, W), dt) for _ in range(3)]
        for y in range(H):
            rgb[band_of(y)][y, :] = ramp
        for a, p in zip(f.planes, (rgb[1], rgb[2], rgb[0])):     # planes are G, B, R
            a[:, :p.shape[1] * p.itemsize] = p.view(np.uint8).reshape(H, -1)
        return f
    f = OL.Frame("rgb24", W, H)
    img = np.zeros((H, W, 3), np.uint8)
    ramp = (256 * np.arange(W) // W).astype(np.uint8)
    for y in range(H):
        img[y, :, band_of(y)] = ramp
    f.planes[0][:, :3 * W] = img.reshape(H, 3 * W)
    return f


def base_picture(base):
    if base == "rgb24":
        return rgbtestsrc()
    if base == "rgb48":   # rgbtest_put_pixel RGB48 (vsrc_testsrc.c:1020-1027): 16-bit ramp c = 65536 * x / w per band;
        # the three words are stored as v16 >> 32, >> 16, >> 0 with r in the LOW word: the r band lands in the third slot
        f = OL.Frame("rgb48le", W, H)
        img = np.zeros((H, W, 3), np.dtype("<u2"))
        ramp = (65536 * np.arange(W) // W).astype(np.uint16)
        for y in range(H):
            img[y, :, 2 - band_of(y)] = ramp
        f.planes[0][:, :6 * W] = img.view(np.uint8).reshape(H, 6 * W)
        return f
    if base.startswith("gbrp"):
        return rgbtestsrc({"gbrp": 8, "gbrp10": 10, "gbrp12": 12, "gbrp16": 16}[base])
    if base in ("nv24", "p410", "p412", "p416"):
        return yuvtestsrc({"nv24": 8, "p410": 10, "p412": 12, "p416": 16}[base], semi=True)
    return yuvtestsrc({"yuv444p": 8, "yuv444p10": 10, "yuv444p12": 12, "yuv444p16": 16}[base])


def cases():
    """[(golden key, base or None, middle format)]"""
    out = []
    for key in sorted(GOLDEN):
        base, _, name = key.rpartition("-")
        out.append((key, base or None, FMT_OF.get(name, name)))
    return out


def fate_pixfmt_md5(key, base, fmt, convert):
    """run one fate-pixfmt pipeline; convert(src_frame, src_fmt, dst_fmt, dither_none) -> dst frame (host)."""
    if base is None:      # pixfmt_conversion: vsynth1 -> fmt -> yuv444p, one frame
        mid = vsynth_frame() if fmt == "yuv420p" else convert(vsynth_frame(), "yuv420p", fmt, False)
        out = mid if fmt == "yuv444p" else convert(mid, fmt, "yuv444p", False)
        reps = 1
    else:                 # pixfmt_conversion_ext: testsrc(base) -> fmt (sws_dither=none) -> base, 25 frames at -t 1
        bf = BASE_FMT[base]
        src = base_picture(base)
        out = src if fmt == bf else convert(convert(src, bf, fmt, True), fmt, bf, False)
        reps = 25
    data = out.visible() * reps
    assert len(data) == GOLDEN[key]["bytes"]
    return hashlib.md5(data).hexdigest()
