"""Shared by tests/test_oracle_fate_nut.py and tests/test_gpu_fate_nut.py: what FATE's `pixfmts` recipe (tests/fate-run.sh:621-660) feeds the NUT
muxer for one pixel format F.

    ffmpeg -f image2 -vcodec pgmyuv -i tests/vsynth1/%02d.pgm -flags +bitexact -sws_flags +accurate_rnd+bitexact -fflags +bitexact
           -vf "scale,format=F,<filter>" -vcodec rawvideo -pix_fmt F -frames:v 1 -f nut md5:

  copy / null   <filter> passes the picture on: the stream holds vsynth1 frame 0 (352x288 yuv420p) converted to F by the first `scale`;
  scale         <filter> = scale=200:100: that picture scaled from 352x288 to 200x100 inside F.
Both conversions are libavfilter/vf_scale.c:scale_frame (:779-866) -> sws_scale_frame() on a context that carries only the flags (bicubic from
the option's default + accurate_rnd + bitexact); everything else comes from the frames' properties, all of them unspecified here (a pgmyuv
picture).  A pal8 output is produced as bgr8 with the systematic palette (vf_scale.c:847-850, imgutils.c:191-195) and stored with its 1024-byte
palette behind the indices (av_image_copy_to_buffer); yuvj* are their full-range twins (format.c:330-336)."""
import json
import os
import zlib

import numpy as np

import nut_mux
import oracle_lib as OL

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "fate_nut_md5.json")))
W, H = 352, 288
FLAGS = OL.SWS_BICUBIC | OL.SWS_ACCURATE_RND | OL.SWS_BITEXACT
NAME = {"gray": "gray8"}                      # pixdesc name -> the name tests/oracle_lib.py uses


def lib_name(fmt):
    return NAME.get(fmt, fmt)


def vsynth_frame0():
    raw = zlib.decompress(open(os.path.join(HERE, "golden", "vsynth1_f0_352x288.yuv420p.bin.z"), "rb").read())
    f = OL.Frame("yuv420p", W, H)
    o = 0
    for pl, (rb, rows) in zip(f.planes, ((W, H), (W // 2, H // 2), (W // 2, H // 2))):
        pl[:, :rb] = np.frombuffer(raw, np.uint8, rb * rows, o).reshape(rows, rb)
        o += rb * rows
    return f


def systematic_pal_bgr8():
    pal = bytearray()
    for i in range(256):
        b, g, r = (i >> 6) * 85, ((i >> 3) & 7) * 36, (i & 7) * 36
        pal += (b | g << 8 | r << 16 | 0xFF << 24).to_bytes(4, "little")
    return bytes(pal)


def pack(frame, fmt):
    """av_image_copy_to_buffer(..., align 1): visible rows of every plane back to back (+ the palette of a pal8 picture)"""
    out = bytearray()
    for pl, rb in zip(frame.planes, frame.row_bytes):
        out += np.ascontiguousarray(pl[:, :rb]).tobytes()
    if fmt == "pal8":
        out += systematic_pal_bgr8()
    return bytes(out)


def md5_of(fmt, test, convert):
    """convert(src_frame, src_fmt, dst_fmt, dw, dh) -> frame in dst_fmt: one sws_scale_frame() of vf_scale"""
    work = "bgr8" if fmt == "pal8" else lib_name(fmt)
    pic = vsynth_frame0()
    if work != "yuv420p":                                  # (yuv420p -> yuv420p at the same size: sws_is_noop, the filter passes the frame on)
        pic = convert(pic, "yuv420p", work, W, H)
    w, h = W, H
    if test == "scale":
        src_fmt = "pal8" if fmt == "pal8" else work        # the bgr8 picture travels on as pal8 (vf_scale.c:855) and is read through its palette
        pic = convert(pic, src_fmt, work, 200, 100)
        w, h = 200, 100
    md5, _ = nut_mux.nut_md5([pack(pic, fmt)], w, h, GOLDEN["tags"][fmt])
    return md5


# ---- the yuv420p recipes of tests/fate/filter-video.mak:508-527, :545-546 (video_filter(): five frames of vsynth1, -frames:v 5) ----
VIDEO_FILTER_MD5 = {      # tests/ref/fate/filter-<name>
    "null": "fcb007249fba9371fe84a61c974fcb00", "scale200": "e7b8419c7de2912f0585b79e99f174c2", "scale500": "e7d6f07710a707e4e5583aee54a8f5ff",
    "crop": "59c225f4cdab05af984dd259f10be762", "crop_scale": "728fa480f1b959cddd3f83c92d8719c4", "crop_vflip": "0652fe087e7a0cc110c3a876543b8662",
    "vflip": "0de640dff4447bd1b33d23f2b8ad9d4a", "crop_scale_vflip": "d6a0bb35b159aa6787add0082088a59f",
}


def vsynth_frames(n=5):
    import lzma
    fsz = W * H * 3 // 2
    raw = zlib.decompress(open(os.path.join(HERE, "golden", "vsynth1_f0_352x288.yuv420p.bin.z"), "rb").read())
    raw += lzma.decompress(open(os.path.join(HERE, "golden", "vsynth1_f1-3_352x288.yuv420p.bin.xz"), "rb").read())
    raw += lzma.decompress(open(os.path.join(HERE, "golden", "vsynth1_f4_352x288.yuv420p.bin.xz"), "rb").read())
    out = []
    for k in range(n):
        o = k * fsz
        y = np.frombuffer(raw, np.uint8, W * H, o).reshape(H, W)
        u = np.frombuffer(raw, np.uint8, W * H // 4, o + W * H).reshape(H // 2, W // 2)
        v = np.frombuffer(raw, np.uint8, W * H // 4, o + W * H * 5 // 4).reshape(H // 2, W // 2)
        out.append([y, u, v])
    return out


def _frame(planes):
    h, w = planes[0].shape
    f = OL.Frame("yuv420p", w, h)
    for pl, a in zip(f.planes, planes):
        pl[:a.shape[0], :a.shape[1]] = a
    return f


def _planes(frame):
    return [np.array(pl[:, :rb]) for pl, rb in zip(frame.planes, frame.row_bytes)]


def _crop(p, w, h, x, y):       # libavfilter/vf_crop.c: the plane pointers move by (x, y), chroma by the subsampled offsets
    return [p[0][y:y + h, x:x + w], p[1][y // 2:y // 2 + (h + 1) // 2, x // 2:x // 2 + (w + 1) // 2], p[2][y // 2:y // 2 + (h + 1) // 2, x // 2:x // 2 + (w + 1) // 2]]


def _vflip(p):                  # libavfilter/vf_vflip.c: the last row becomes data[], the line sizes change sign
    return [a[::-1] for a in p]


def video_filter_md5(name, convert):
    """convert(frame, "yuv420p", "yuv420p", dw, dh) -> frame: one scale filter instance (sws_scale_frame; equal sizes are passed through)"""
    def scale(p, w, h):
        if p[0].shape == (h, w):
            return p            # sws_is_noop (vf_scale.c:840-845)
        return _planes(convert(_frame(p), "yuv420p", "yuv420p", w, h))
    frames = []
    for p in vsynth_frames(5):
        if name == "scale200": p = scale(p, 200, 200)
        elif name == "scale500": p = scale(p, 500, 500)
        elif name == "crop": p = _crop(p, W - 100, H - 100, 100, 100)
        elif name == "vflip": p = _vflip(p)
        elif name == "crop_vflip": p = _vflip(_crop(p, W - 100, H - 100, 100, 100))
        elif name == "crop_scale":              # scale=w=400:h=-1: h = av_rescale(400, 188, 252) = 298 (vf_scale's ff_scale_adjust_dimensions)
            p = scale(_crop(p, W - 100, H - 100, 100, 100), 400, 298)
        elif name == "crop_scale_vflip":
            p = _crop(p, W - 200, H - 200, 200, 200)
            p = _crop(p, p[0].shape[1] - 20, p[0].shape[0] - 20, 20, 20)
            p = scale(p, 200, 200); p = scale(p, 250, 250)
            p = _vflip(_vflip(p))
            p = scale(p, 200, 200)
            p = _crop(p, 100, 100, 100, 100)
            p = _vflip(p)
            p = scale(p, 200, 200)
            p = _vflip(p)
            p = _crop(p, 100, 100, 100, 100)
        frames.append(p)
    h, w = frames[0][0].shape
    raw = [b"".join(np.ascontiguousarray(a).tobytes() for a in p) for p in frames]
    return nut_mux.nut_md5(raw, w, h, GOLDEN["tags"]["yuv420p"])[0]
