"""-m gpu parity for the scaled packed-RGB source path (main:rgbread+strip_march): the reader pre-pass sws_k_rgb_read16 (rgb24ToY_c /
rgb24ToUV_half_c, input.c:1068-1172; the 32-bit rows of rgb16_32To*_c_template, :264-393) writes the 16-bit planes hScale16To15_c
reads (swscale.c:99-125, sh = 13 for RGB sources), the marching strip kernel scales them like a planar 16-bit source."""
import numpy as np
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SPLINE, SWS_FULL_CHR_H_INP)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
PATH = "main:rgbread+strip_march"
TUNE = dict(strip_min_w=0, no_strip_rgbsrc=1)     # (no_strip_rgbsrc: this file is about the two-pass form; the one-launch form has test_gpu_strip_rgbsrc.py)
TWO = dict(no_strip_rgbsrc=1)
# (the planner keeps pictures narrower than 320 columns on the tile kernel: force the path onto oracle-sized cases)

SRC = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr", "gbrp", "gbrap"]   # (planar 8-bit GBR: the same readers, three planes)
DST = ["yuv420p", "yuv422p", "yuv444p", "nv12", "nv21", "yuv420p10le", "p010le", "yuv422p12le"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh) in ((256, 64, 192, 48), (320, 50, 512, 80), (132, 33, 66, 17)):
        path, _ = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=TUNE)
        assert path == PATH, (path, sw, sh, dw, dh)


@pytest.mark.parametrize("flags", [SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND],
                         ids=["area", "bilinear", "bicubic", "gauss", "lanczos", "spline", "accurate"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (640, 96, 426, 64), (260, 200, 520, 300), (1924, 34, 1282, 22), (64, 40, 1030, 44), (2052, 20, 1026, 10)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for src, dst in (("rgb24", "yuv420p"), ("bgra", "nv12"), ("argb", "yuv422p10le")):
        run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=7, tune=TUNE)


def test_planner_and_fallbacks():
    assert run_case(1920, 54, "rgb24", 1280, 36, "yuv420p", SWS_BICUBIC | BX, tune=TWO)[0] == PATH                       # wide enough without the option
    assert run_case(480, 48, "rgb24", 240, 24, "yuv420p", SWS_BICUBIC | BX, tune=TWO)[0] != PATH                         # narrow: tile kernel
    assert run_case(642, 48, "rgb24", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0] == PATH              # a width of 4 k + 2 (round 5: the last group of four pixels is read whole)
    assert run_case(643, 48, "rgb24", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0] != PATH              # odd width
    assert run_case(640, 48, "rgb24", 480, 36, "yuv420p", SWS_BILINEAR | SWS_FULL_CHR_H_INP | BX, tune=TUNE)[0] == PATH            # the full-width chroma readers
    assert run_case(640, 48, "rgb24", 320, 24, "yuv420p", SWS_BICUBIC | SWS_FULL_CHR_H_INP | BX, tune=TUNE)[0] == PATH             # (4:1 bicubic chroma, 17 taps: the strip kernel's long form)
    assert run_case(640, 48, "rgba", 320, 24, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0] != PATH
    assert run_case(640, 48, "rgb24", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=TUNE)[0] == PATH             # a range conversion (round 5: converted on the way into the ring)
    assert run_case(640, 48, "rgb24", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_range=1))[0] != PATH
    assert run_case(640, 48, "rgb24", 320, 24, "yuv420p", SWS_POINT | BX, tune=TUNE)[0] is not None


def test_full_size_frames_and_host_frames():
    assert run_case(1920, 1080, "rgb24", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2, tune=TWO)[0] == PATH
    assert run_case(1920, 1080, "bgra", 2560, 1440, "nv12", SWS_BILINEAR | BX, seed=3, tune=TWO)[0] == PATH                                  # (chroma wider than half the source: full-width readers)
    assert run_case(1920, 1080, "bgra", 3840, 2160, "nv12", SWS_BILINEAR | BX, seed=3, tune=TWO)[0] == PATH                                  # (2x: the chroma planes are not scaled at all: one-tap filters)
    assert run_case(2560, 1440, "rgb24", 1920, 1080, "yuv420p10le", SWS_LANCZOS | BX, seed=4, device_frames=False, tune=TWO)[0] == PATH


def test_batches():
    """several frames per sws_scale_frames() call: every frame has its own reader planes, found through the second frame table"""
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    for src, dst, sw, sh, dw, dh, n, flags in (("rgb24", "yuv420p", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("bgra", "nv12", 1024, 130, 1536, 190, 3, SWS_LANCZOS | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
        p.set_option("no_strip_rgbsrc", 1)
        refs, srcs, dsts = [], [], []
        for k in range(n):
            s = OL.fill_random(OL.Frame(src, sw, sh), 40 + k)
            ref = OL.Frame(dst, dw, dh)
            assert o.scale(s, ref) == dh
            refs.append(ref)
            hs = HostFrame(src, sw, sh)
            for a, b in zip(hs.planes, s.planes):
                a[:] = b
            srcs.append(DeviceFrame(src, sw, sh).upload(hs))
            dsts.append(DeviceFrame(dst, dw, dh))
        torch.cuda.synchronize()
        for rep in range(2):     # (the second call finds the cached frame table)
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
            assert p.path() == PATH
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)


@pytest.mark.parametrize("src", ["rgb24", "bgr24", "bgra", "rgba", "argb", "rgb0", "gbrp", "gbrap", "x2rgb10le", "rgb565le", "gbrp10le", "rgb48le", "rgba64le", "gbrpf32le"])
def test_widths_of_4k_plus_2(src):
    """1366 x 768 screens, 854 x 480: source widths that are even but not multiples of 4 through the reader pre-pass (the last group of four pixels is read and written whole;
    the one-launch kernels keep their multiple-of-4 rule and hand such pictures to the pre-pass), into YUV, gray and RGB destinations, from HBM frames with padded rows and from
    host frames"""
    for dst in ("yuv420p", "nv12", "yuv444p", "yuv420p10le", "bgra", "rgb24", "gray8", "yuyv422", "yuva420p", "yuv420p16le", "rgba", "yuva444p", "gbrap"):
        for (sw, sh, dw, dh, fl) in ((1366, 48, 1280, 44, SWS_BICUBIC), (854, 48, 1282, 72, SWS_BICUBIC), (642, 30, 322, 15, SWS_BILINEAR), (1918, 22, 1278, 14, SWS_LANCZOS), (646, 26, 646, 26, SWS_BICUBIC),
                                     (650, 33, 400, 33, SWS_BILINEAR)):
            run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh)
    # (rows without room behind their last pixel: 170 x 3 = 510 bytes in a line of 512, 426 x 3 = 1278 in 1280 -- the last group's loads stop at the last pixel)
    for dst in ("rgb0", "gbrp10le", "argb", "yuv420p", "bgr444le"):
        for (sw, sh, dw, dh, fl) in ((170, 77, 608, 63, SWS_BICUBIC), (170, 20, 161, 26, SWS_BICUBIC), (170, 23, 334, 28, SWS_BILINEAR), (426, 78, 580, 79, SWS_BICUBIC)):
            run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=dict(strip_min_w=0))
    assert "rgbread" in run_case(1366, 48, src, 1280, 44, "yuv420p", SWS_BICUBIC | BX, seed=1)[0]
    run_case(1366, 768, src, 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2, device_frames=False)
    run_case(1366, 768, src, 1280, 720, "bgra", SWS_BILINEAR, seed=3)
