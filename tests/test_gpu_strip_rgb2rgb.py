"""-m gpu parity for the one-launch form of scaled packed RGB -> packed RGB (main:strip_rgb2rgb, sws_k_strip_rgb2rgb: kernels_striprgb2rgb.hpp):
readers (rgb24ToY_c / rgb24ToUV_c / rgb24ToUV_half_c, the 32-bit rows of rgb16_32To*_c_template, rgbaToA_c / abgrToA_c), hScale16To15_c of Y, U, V
and A, the vertical filters and yuv2rgb_full_X_c_template + yuv2rgb_write_full in one wave.  Every case is compared with the oracle byte for
byte; the helper-pass form (reader pre-pass + strip launches + sws_k_fullchr_rgb) keeps its tests in test_gpu_fullchr.py / test_gpu_strip_kernel.py."""
import numpy as np
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SPLINE, SWS_SINC, SWS_FULL_CHR_H_INP, SWS_FULL_CHR_H_INT)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
PATH = "main:strip_rgb2rgb"
OLD = "main:rgbread+strip_march+fullchr_rgb"
TUNE = dict(strip_min_w=0)

SRC = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr", "bgr0", "0rgb"]
DST = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    # down by more than 2 (the half chroma readers), down by less, up, one axis only
    for (sw, sh, dw, dh) in ((256, 64, 192, 48), (320, 50, 512, 80), (264, 66, 128, 32), (644, 38, 322, 19), (256, 40, 256, 64), (400, 40, 300, 40)):
        r = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw + dh, tune=TUNE)
        assert r[0] == PATH, (r[0], src, dst, sw, sh, dw, dh)


@pytest.mark.parametrize("flags", [SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_SINC, SWS_POINT, SWS_BICUBIC | SWS_ACCURATE_RND,
                                   SWS_BICUBIC | SWS_FULL_CHR_H_INP, SWS_BILINEAR | SWS_FULL_CHR_H_INT],
                         ids=["area", "bilinear", "bicubic", "gauss", "lanczos", "spline", "sinc", "point", "accurate", "chr_h_inp", "chr_h_int"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (640, 96, 426, 64), (260, 200, 520, 300), (1924, 34, 1282, 22), (64, 40, 1030, 44), (2052, 20, 1026, 10),
                                  (1280, 90, 854, 61), (700, 301, 333, 97), (640, 64, 640, 32), (640, 64, 320, 64), (1024, 37, 1024, 111), (960, 300, 512, 80), (250, 33, 131, 17)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for src, dst in (("rgb24", "rgb24"), ("bgra", "bgra"), ("argb", "bgr24"), ("bgr24", "rgba"), ("rgb0", "abgr")):
        run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=11, tune=TUNE)


def test_planner_and_fallbacks():
    assert run_case(1920, 54, "bgra", 1280, 36, "bgra", SWS_BICUBIC | BX)[0] == PATH
    assert run_case(1920, 54, "bgra", 1280, 36, "bgra", SWS_BICUBIC | BX, tune=dict(no_strip_rgb2rgb=1))[0] == OLD
    assert run_case(1282, 48, "rgb24", 642, 24, "rgb24", SWS_BICUBIC | BX, tune=TUNE)[0] == OLD                # source width not a multiple of 4: the reader pre-pass (which takes 4 k + 2), strip launches, epilogue
    assert run_case(1280, 96, "rgb24", 320, 24, "rgb24", SWS_BICUBIC | BX, tune=TUNE)[0] != PATH               # 17 taps: the long forms
    assert run_case(640, 48, "gbrp", 480, 36, "rgb24", SWS_BILINEAR | BX, tune=TUNE)[0] != PATH                # planar RGB source
    assert run_case(640, 48, "rgb24", 480, 36, "gbrp", SWS_BILINEAR | BX, tune=TUNE)[0] != PATH                # planar RGB destination: its own epilogue
    assert run_case(640, 48, "rgb24", 480, 36, "rgb565le", SWS_BILINEAR | BX, tune=TUNE)[0] != PATH            # 16 bpp: dithered writers
    assert run_case(640, 48, "rgba", 320, 24, "rgba", SWS_BICUBIC | BX, tune=TUNE, opts=None)[0] == PATH       # alpha through the luma filters


def test_full_size_frames_and_host_frames():
    assert run_case(3840, 2160, "bgra", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=2)[0] == PATH
    assert run_case(1920, 1080, "rgb24", 1280, 720, "rgb24", SWS_BICUBIC | BX, seed=3)[0] == PATH
    assert run_case(1920, 1080, "bgra", 3840, 2160, "bgra", SWS_BICUBIC | BX, seed=4)[0] == PATH
    assert run_case(2560, 1440, "rgba", 1920, 1080, "rgb24", SWS_LANCZOS | BX, seed=5, device_frames=False)[0] == PATH
    assert run_case(1920, 1080, "bgr0", 1280, 720, "bgra", SWS_BILINEAR | BX, seed=6)[0] == PATH


def test_batches():
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    for src, dst, sw, sh, dw, dh, n, flags in (("bgra", "bgra", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("rgb24", "bgra", 1024, 130, 400, 50, 9, SWS_LANCZOS | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
        refs, srcs, dsts = [], [], []
        for k in range(n):
            s = OL.fill_random(OL.Frame(src, sw, sh), 140 + k)
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
            assert p.path() == PATH and p.kernel_name() == "sws_k_strip_rgb2rgb"
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)
        p.close()
