"""-m gpu parity for full-chroma packed RGB destinations through the strip kernels ("+fullchr_rgb"): SWS_FULL_CHR_H_INT -- set by the caller, or forced
for RGB and 4:4:4 sources and odd destination widths (utils.c:1270-1286) -- scales Y, U and V to the destination size; the strip kernels leave their
vertical sums as int32 planes and sws_k_fullchr_rgb finishes yuv2rgb_full_X_c_template / yuv2rgb_write_full (output.c:2005-2070, :2163-2207)."""
import numpy as np
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_AREA, SWS_GAUSS, SWS_FULL_CHR_H_INT)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
FC = SWS_FULL_CHR_H_INT
TUNE = dict(strip_min_w=0)

SRC = ["rgb24", "bgr24", "bgra", "argb", "rgb0", "0bgr", "gbrp", "gbrap", "yuv444p", "yuv420p", "yuv422p", "nv12", "nv21", "yuyv422", "uyvy422", "yuv420p10le", "yuv444p10le", "p010le", "yuvj420p",
       "yuv410p", "yuva420p", "yuva444p", "yuva422p10le"]
SUBSAMPLED_V = ("yuv420p", "nv12", "nv21", "yuv420p10le", "p010le", "yuvj420p", "yuv410p", "yuva420p")
DST = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 67, 17, SWS_AREA), (256, 64, 321, 96, SWS_LANCZOS),
                                 (256, 64, 250, 64, SWS_GAUSS), (130, 30, 131, 31, SWS_BICUBIC | SWS_ACCURATE_RND), (256, 64, 256, 64, SWS_BICUBIC)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | FC | BX, seed=sw + dh, tune=TUNE)
        # (a real alpha plane scaled into a 32 bpp destination -- needAlpha -- is a fourth sum plane: the reader pre-pass hands the A bytes of a packed
        #  source to the luma filters, a planar YUV source has them in plane 3; planar RGB with alpha keeps the generic writer)
        alpha_generic = src == "gbrap" and dst not in ("rgb24", "bgr24")
        # (area at 2:1: two luma and two chroma taps, yuv2rgb_full_2: the generic writer.  An unscaled height with unsubsampled chroma rows -- one tap each,
        #  yuv2rgb_full_1 -- is the X arithmetic with the tap 4096 and takes the strip kernels)
        short = fl == SWS_AREA
        rgb_src = src in ("rgb24", "bgr24", "bgra", "argb", "rgb0", "0bgr", "gbrp", "gbrap")
        if (sw, sh) != (dw, dh) and not alpha_generic and not short and not (rgb_src and sw & 3):
            assert r[0].endswith("+fullchr_rgb") or r[0] == "main:strip_rgb2rgb", (r[0], src, dst, sw, dw)     # (packed RGB -> packed RGB: the one-launch form, test_gpu_strip_rgb2rgb.py)


GBR_DST = ["gbrp", "gbrap", "gbrp9le", "gbrp10le", "gbrp12le", "gbrp14le", "gbrap10le", "gbrap12le", "gbrap14le", "gbrp10msble", "gbrp12msble", "gbrp10be", "gbrap12be"]


@pytest.mark.parametrize("src", ["yuv420p", "nv12", "yuv444p", "yuv420p10le", "p010le", "rgb24", "bgra", "rgb0", "gbrp", "yuva420p", "yuva444p10le", "yuyv422", "yuvj420p"])
@pytest.mark.parametrize("dst", GBR_DST)
def test_planar_rgb_destinations(src, dst):
    """yuv2gbrp_full_X_c (output.c:2342-2421): planar RGB of 8 .. 14 bits always takes the X form and forced full chroma: sws_k_fullchr_gbrp"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BILINEAR), (132, 34, 67, 17, SWS_AREA), (256, 64, 321, 96, SWS_LANCZOS),
                                 (256, 64, 250, 64, SWS_GAUSS), (132, 30, 131, 31, SWS_BICUBIC | SWS_ACCURATE_RND), (256, 64, 256, 32, SWS_BICUBIC)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=TUNE)
        if sw != dw and not (src in ("rgb24", "bgra", "rgb0", "gbrp") and sw & 3):
            assert r[0].endswith("+fullchr_rgb"), (r[0], src, dst, sw, dw)


def test_forced_full_chroma_and_fallbacks():
    assert run_case(3840, 2160, "yuv420p", 1920, 1080, "gbrp", SWS_BICUBIC | BX, seed=9)[0] == "main:strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "nv12", 1280, 720, "gbrap", SWS_BILINEAR | BX, seed=10, device_frames=False)[0] == "main:strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "bgra", 1280, 720, "gbrap10le", SWS_BICUBIC | BX, seed=11)[0] == "main:rgbread+strip_march+fullchr_rgb"
    assert not run_case(256, 64, "yuv420p", 192, 48, "gbrp16le", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")     # 19-bit intermediates
    assert not run_case(256, 64, "yuv420p", 192, 48, "gbrpf32le", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")
    assert run_case(256, 64, "rgb24", 192, 48, "bgr24", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_rgb2rgb=1))[0].endswith("+fullchr_rgb")        # RGB source: forced
    assert run_case(256, 64, "yuv444p", 192, 48, "bgra", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")        # 4:4:4 source: forced
    assert run_case(256, 64, "yuv420p", 191, 48, "rgb24", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")       # odd width: forced
    assert not run_case(256, 64, "yuv420p", 192, 48, "rgb24", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")   # the LUT writers
    assert run_case(256, 64, "bgra", 192, 48, "bgra", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"   # alpha plane scaled: a fourth sum plane
    assert run_case(256, 64, "yuva420p", 192, 48, "rgba", SWS_BICUBIC | FC | BX, tune=TUNE)[0] == "main:strip_march+fullchr_rgb"
    assert not run_case(256, 64, "gbrap", 192, 48, "rgba", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")      # planar RGB with alpha: the generic writer
    assert run_case(256, 64, "yuva420p16le", 192, 48, "rgba", SWS_BICUBIC | FC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")       # 16-bit samples: round 5 (strip_hstage_b)
    assert not run_case(256, 64, "yuva420p16le", 192, 48, "rgba", SWS_BICUBIC | FC | BX, tune=dict(TUNE, no_strip_u16=1))[0].endswith("+fullchr_rgb")
    assert not run_case(256, 64, "rgb24", 192, 48, "rgb565le", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+fullchr_rgb")
    assert run_case(256, 64, "rgb24", 256, 128, "bgr24", SWS_BILINEAR | BX, tune=dict(TUNE, no_strip_rgb2rgb=1))[0].endswith("+fullchr_rgb")     # two luma and two chroma taps: yuv2rgb_full_2 (round 5: the epilogue leaves the rounding out of such rows)
    assert run_case(256, 64, "rgb24", 256, 128, "bgr24", SWS_BILINEAR | BX, tune=TUNE)[0] == "main:strip_rgb2rgb"        # (... and so does the one-launch kernel, from its row entries)
    assert not run_case(256, 64, "rgb24", 256, 128, "bgr24", SWS_BILINEAR | BX, tune=dict(TUNE, no_short_forms=1))[0].endswith("+fullchr_rgb")
    assert not run_case(480, 48, "rgb24", 240, 24, "bgr24", SWS_BICUBIC | BX)[0].endswith("+fullchr_rgb")                # narrow: below the planner's width threshold


def test_one_tap_vertical_forms():
    """both vertical filters with one tap: yuv2rgb_full_1 / yuv2rgb_1 ignore the coefficient (4095 in some rows of an error-diffused bank); same size (every filter the
    identity: yuv444p -> bgra has no unscaled converter) and width-only scaling"""
    for src in ("yuv444p", "yuv444p10le", "yuv422p", "yuvj444p", "rgb24", "bgra", "gbrp", "yuva444p", "nv24", "yuv440p10le"):
        for dst in ("bgra", "rgb24", "argb", "bgr24", "gbrp", "gbrap"):
            for (sw, sh, dw, dh, fl) in ((256, 64, 256, 64, SWS_BICUBIC), (320, 50, 200, 50, SWS_BICUBIC), (200, 37, 320, 37, SWS_LANCZOS), (1920, 24, 1920, 24, SWS_BILINEAR),
                                         (1920, 24, 480, 24, SWS_BICUBIC)):
                r = run_case(sw, sh, src, dw, dh, dst, fl | FC | BX, seed=sw + dh, tune=TUNE)
    # (a 4:4:4 planar source at the same size: the epilogue reads the source planes itself)
    assert run_case(1920, 1080, "yuv444p", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=41)[0] == "main:fullchr_rgb_direct"
    assert run_case(1920, 1080, "yuv444p10le", 1920, 1080, "rgb24", SWS_BICUBIC | BX, seed=42)[0] == "main:fullchr_rgb_direct"
    assert run_case(1920, 1080, "yuva444p", 1920, 1080, "rgba", SWS_BICUBIC | BX, seed=45, device_frames=False)[0] == "main:fullchr_rgb_direct"
    assert run_case(1918, 1080, "yuv444p12le", 1918, 1080, "gbrp12le", SWS_BICUBIC | BX, seed=46)[0] == "main:fullchr_rgb_direct"
    assert run_case(1921, 270, "yuvj444p", 1921, 270, "gbrap", SWS_BICUBIC | BX, seed=47)[0] == "main:fullchr_rgb_direct"
    for src in ("yuv444p", "yuv444p9le", "yuv444p10le", "yuv444p14le", "yuva444p", "yuva444p10le", "yuvj444p"):
        for dst in ("rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "gbrp", "gbrap", "gbrp10le", "gbrap12le", "gbrp12msble"):
            for (w, h) in ((256, 64), (322, 50), (129, 33), (67, 18), (1026, 21), (1, 1), (3, 2), (5, 3)):
                run_case(w, h, src, w, h, dst, SWS_BICUBIC | BX, seed=w + h, tune=TUNE)
    assert run_case(1920, 1080, "yuv444p16le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=48)[0] != "main:fullchr_rgb_direct"   # 16-bit samples
    assert run_case(1920, 1080, "rgb24", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=43, tune=dict(no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"   # (BITEXACT: no rgb24 -> bgra shuffle, swscale_unscaled.c findRgbConvFn)
    assert run_case(1920, 1080, "yuv444p", 1280, 1080, "bgra", SWS_BICUBIC | BX, seed=44)[0] == "main:strip_march+fullchr_rgb"


def test_full_size_batches_and_host_frames():
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    assert run_case(3840, 2160, "rgb24", 1920, 1080, "rgb24", SWS_BICUBIC | BX, seed=2, tune=dict(no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "bgr24", 1280, 720, "bgra", SWS_BILINEAR | BX, seed=3, device_frames=False, tune=dict(no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "yuv444p", 1280, 720, "rgb24", SWS_BICUBIC | BX, seed=4)[0] == "main:strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "yuv420p", 1280, 720, "bgra", SWS_LANCZOS | FC | SWS_ACCURATE_RND | BX, seed=5)[0] == "main:strip_march+fullchr_rgb"
    assert run_case(3840, 2160, "bgra", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=6, tune=dict(no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "rgba", 1280, 720, "argb", SWS_LANCZOS | BX, seed=7, device_frames=False, tune=dict(no_strip_rgb2rgb=1))[0] == "main:rgbread+strip_march+fullchr_rgb"
    assert run_case(1920, 1080, "yuva420p", 1280, 720, "bgra", SWS_BICUBIC | FC | BX, seed=8)[0] == "main:strip_march+fullchr_rgb"
    for src, dst, sw, sh, dw, dh, n, flags in (("rgb24", "bgr24", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("nv12", "bgra", 1024, 64, 1283, 80, 3, SWS_BICUBIC | FC | BX),
                                               ("abgr", "rgba", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("yuva444p10le", "bgra", 1024, 64, 1283, 80, 3, SWS_BICUBIC | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
        p.set_option("no_strip_rgb2rgb", 1)        # (this file is about the helper-pass form)
        refs, srcs, dsts = [], [], []
        for k in range(n):
            s = OL.fill_random(OL.Frame(src, sw, sh), 60 + k)
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
            assert p.path().endswith("+fullchr_rgb"), p.path()
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)
