"""-m gpu parity for the one-launch form of scaled packed-RGB sources (main:strip_rgbsrc, sws_k_strip_rgbsrc: kernels_striprgbsrc.hpp): the
readers (rgb24ToY_c / rgb24ToUV_half_c, input.c:1068-1172; the 32-bit rows of rgb16_32To*_c_template, :264-393), hScale16To15_c for luma and
chroma, the vertical filters and the planar / semi-planar writers in one wave -- luma and chroma marching in lockstep over the source row pairs.
Every case is compared with the oracle byte for byte; the two-pass form (reader pre-pass + two strip launches) has test_gpu_rgbread.py."""
import numpy as np
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SPLINE, SWS_SINC, SWS_FULL_CHR_H_INP)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
PATH = "main:strip_rgbsrc"
TUNE = dict(strip_min_w=0)     # (the planner keeps pictures narrower than 320 columns on the tile kernel: force the path onto oracle-sized cases)

SRC = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr", "gbrp", "gbrap",    # (planar 8-bit GBR: the same readers, three planes)
       "x2rgb10le", "x2bgr10le"]                                                             # (round 5: the rgb30le / bgr30le rows of rgb16_32To*_c_template, input.c:411-412)
DST = ["yuv420p", "yuv422p", "nv12", "nv21", "yuv420p10le", "p010le", "yuv422p12le", "yuv420p16le", "yuv420p9be", "nv16", "p012be"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh) in ((256, 64, 192, 48), (320, 50, 512, 80), (132, 33, 66, 17), (644, 38, 322, 19)):
        r = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=TUNE)
        # (up-scaling: the chroma planes are wider than half the source, the full-width chroma readers, the two-pass form; formats outside this list: whatever the planner gives them)
        if dw <= sw and dst in ("yuv420p", "yuv422p", "nv12", "nv21", "yuv420p10le", "p010le", "yuv422p12le"):
            assert r[0] == PATH, (r[0], sw, sh, dw, dh)


@pytest.mark.parametrize("flags", [SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_SINC, SWS_POINT, SWS_BICUBIC | SWS_ACCURATE_RND],
                         ids=["area", "bilinear", "bicubic", "gauss", "lanczos", "spline", "sinc", "point", "accurate"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (640, 96, 426, 64), (260, 200, 520, 300), (1924, 34, 1282, 22), (64, 40, 1030, 44), (2052, 20, 1026, 10),
                                  (1280, 90, 854, 61), (700, 301, 333, 97), (640, 64, 640, 32), (640, 64, 320, 64), (1024, 37, 1024, 111), (960, 300, 512, 80)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for src, dst in (("rgb24", "yuv420p"), ("bgra", "nv12"), ("argb", "yuv422p10le"), ("bgr24", "p010le"), ("gbrp", "yuv420p"), ("x2rgb10le", "p010le"), ("x2bgr10le", "yuv420p")):
        run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=7, tune=TUNE)


def test_planner_and_fallbacks():
    assert run_case(1920, 54, "rgb24", 1280, 36, "yuv420p", SWS_BICUBIC | BX)[0] == PATH                       # wide enough without the option
    assert run_case(1920, 54, "rgb24", 1280, 36, "yuv420p", SWS_BICUBIC | BX, tune=dict(no_strip_rgbsrc=1))[0] == "main:rgbread+strip_march"
    assert run_case(480, 48, "rgb24", 240, 24, "yuv420p", SWS_BICUBIC | BX)[0] != PATH                         # narrow: tile kernel
    assert run_case(642, 48, "rgb24", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:rgbread+strip_march"    # width not a multiple of 4: the reader pre-pass (which takes 4 k + 2)
    assert run_case(640, 48, "rgb24", 480, 36, "yuv420p", SWS_BILINEAR | SWS_FULL_CHR_H_INP | BX, tune=TUNE)[0] == "main:rgbread+strip_march"   # the full-width chroma readers
    assert run_case(640, 48, "rgb24", 480, 36, "yuv444p", SWS_BILINEAR | BX, tune=TUNE)[0] == "main:rgbread+strip_march"    # full-width chroma planes
    assert run_case(640, 48, "gbrp", 480, 36, "yuv420p", SWS_BILINEAR | BX, tune=TUNE)[0] == PATH                           # planar RGB: three planes, the same readers
    assert run_case(640, 48, "gbrp10le", 480, 36, "yuv420p", SWS_BILINEAR | BX, tune=TUNE)[0] != PATH                       # (deeper planar RGB: other readers)
    assert run_case(640, 48, "rgba", 320, 24, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:rgbread+strip_march+alpha"
    assert run_case(1280, 96, "rgb24", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:rgbread+strip_march"    # 17 taps: the long forms
    assert run_case(640, 48, "rgb24", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=TUNE)[0] == PATH             # a range conversion (round 5: converted on the way into the rings)
    assert run_case(640, 48, "rgb24", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_range=1))[0] != PATH


def test_x2rgb10_extremes():
    """all-ones / all-zeros / X bits set / single-field pictures through the 30 bpp reader: the 32-bit wrap-around sums of the reference
    (input.c:283-294, :348-366) and the pair sums' carries into the neighbouring field"""
    import oracle_lib as OL
    for src in ("x2rgb10le", "x2bgr10le"):
        for k, word in enumerate((0xFFFFFFFF, 0x00000000, 0xC0000000, 0x3FF00000, 0x000FFC00, 0x000003FF, 0x3FFFFFFF, 0xEAAAAAAA, 0x95555555)):
            f = OL.Frame(src, 640, 48)
            f.planes[0].view(np.uint32)[:] = word
            if k >= 7:
                f.planes[0].view(np.uint32)[:, ::2] ^= 0xFFFFFFFF
            for dst, dw, dh in (("yuv420p", 320, 24), ("p010le", 640, 48), ("yuvj420p", 480, 36)):
                assert run_case(640, 48, src, dw, dh, dst, SWS_BICUBIC | BX, tune=TUNE, source=f)[0] == PATH


def test_full_size_frames_and_host_frames():
    assert run_case(3840, 2160, "x2rgb10le", 3840, 2160, "p010le", SWS_BICUBIC | BX, seed=12)[0] == PATH       # (HDR desktop capture into a 10-bit encoder)
    assert run_case(3840, 2160, "x2bgr10le", 1920, 1080, "yuv420p10le", SWS_BILINEAR | BX, seed=13)[0] == PATH
    assert run_case(1920, 1080, "rgb24", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2)[0] == PATH
    assert run_case(3840, 2160, "bgra", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=3)[0] == PATH            # (chroma: a 4:1 vertical step, 17 taps: the ring of 12 row pairs)
    assert run_case(3840, 2160, "bgra", 1920, 1080, "nv12", SWS_BILINEAR | BX, seed=5)[0] == PATH
    assert run_case(1920, 1080, "bgra", 1280, 720, "p010le", SWS_LANCZOS | BX, seed=6)[0] == PATH
    assert run_case(1920, 1080, "gbrp", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=9)[0] == PATH
    assert run_case(2560, 1440, "rgb24", 1920, 1080, "yuv420p10le", SWS_LANCZOS | BX, seed=4, device_frames=False)[0] == PATH
    assert run_case(1280, 720, "bgra", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=8)[0] in (PATH, "main:rgbread+strip_march")   # (up: chroma wider than half the source takes the full-width readers)


def test_batches():
    """several frames per sws_scale_frames() call, twice (the second call finds the cached frame table)"""
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    for src, dst, sw, sh, dw, dh, n, flags in (("rgb24", "yuv420p", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("bgra", "nv12", 1024, 130, 768, 96, 9, SWS_LANCZOS | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
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
        for rep in range(2):
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
            assert p.path() == PATH and p.kernel_name() == "sws_k_strip_rgbsrc"
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)
        p.close()
