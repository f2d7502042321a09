"""-m gpu parity for scaled alpha planes behind the strip kernels: a source with alpha into a planar YUV destination with alpha (needAlpha, utils.c:1746)
-- the reference scales the A samples with the luma filters (swscale.c:478-486: alpPixBuf through hyscale and the luma vertical filter, written by the
luma plane's writer into dst[3], vscale.c:66-70).  A packed 32 bpp source hands its A bytes over from the reader pre-pass (rgbaToA_c / abgrToA_c,
input.c:454-472: a << 6 | a >> 2), a planar YUV source has them in plane 3; path suffix "+alpha"."""
import numpy as np
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_AREA, SWS_GAUSS, SWS_POINT
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
TUNE = dict(strip_min_w=0)

SRC = ["bgra", "rgba", "argb", "abgr", "rgb0", "0bgr", "yuva420p", "yuva422p", "yuva444p", "yuva420p10le", "yuva444p10le", "yuva422p12le"]
DST = ["yuva420p", "yuva422p", "yuva444p", "yuva420p10le", "yuva444p10le", "yuva422p12le", "yuva420p9le"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 67, 17, SWS_AREA), (256, 64, 321, 96, SWS_LANCZOS),
                                 (256, 64, 250, 64, SWS_GAUSS), (132, 30, 131, 31, SWS_BICUBIC | SWS_ACCURATE_RND), (256, 64, 256, 32, SWS_BILINEAR),
                                 (256, 64, 128, 128, SWS_POINT)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=TUNE)
        if sw != dw:
            assert r[0].endswith("+alpha"), (r[0], src, dst, sw, dw)


@pytest.mark.parametrize("src", ["yuva420p", "yuva422p", "yuva420p10le", "yuva422p12le", "yuva420p9le"])
@pytest.mark.parametrize("dst", ["rgba", "bgra", "argb", "abgr", "rgb0", "0bgr"])
def test_lut_writers_with_alpha(src, dst):
    """yuva420p -> bgra without full chroma (yuv2rgba32_X_c / yuv2rgba32_1_X_c ...: the LUT writers with hasAlpha): sws_k_strip_rgb + the alpha launch
    with the raw writer + sws_k_alpha_merge32"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 66, 17, SWS_AREA), (256, 64, 322, 96, SWS_LANCZOS),
                                 (256, 64, 250, 64, SWS_GAUSS), (132, 30, 130, 31, SWS_BICUBIC | SWS_ACCURATE_RND), (256, 64, 256, 32, SWS_BILINEAR),
                                 (256, 64, 128, 128, SWS_POINT), (256, 64, 256, 64, SWS_BICUBIC)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=TUNE)
        if (sw, sh, dw, dh, fl) == (256, 64, 192, 48, SWS_BICUBIC):
            assert r[0] == "main:strip_rgb+alpha", (r[0], src, dst)


def test_range_conversion_and_fallbacks():
    assert run_case(1920, 1080, "yuva420p", 1280, 720, "bgra", SWS_BICUBIC | BX, seed=12)[0] == "main:strip_rgb+alpha"
    assert run_case(3840, 2160, "yuva420p10le", 1920, 1080, "rgba", SWS_LANCZOS | BX, seed=13, device_frames=False)[0] == "main:strip_rgb+alpha"
    # (a luma range conversion never touches the alpha plane -- hscale.c:61-63 vs :66-79 -- the alpha launch runs with the conversion switched off)
    ro = dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
    assert run_case(256, 64, "yuva420p", 192, 48, "yuva444p10le", SWS_BICUBIC | BX, tune=TUNE, opts=ro)[0] == "main:strip_march+alpha"
    assert run_case(256, 64, "bgra", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE, opts=ro)[0] == "main:rgbread+strip_march+alpha"
    assert run_case(256, 64, "bgra", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:rgbread+strip_march+alpha"
    assert run_case(256, 64, "yuva420p", 192, 48, "yuva444p10le", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:strip_march+alpha"
    assert run_case(254, 64, "bgra", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")             # (round 5: the reader pre-pass takes widths of 4 n + 2 too)
    assert not run_case(253, 64, "bgra", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")         # odd widths: not
    assert not run_case(256, 64, "gbrap", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")        # planar RGB with alpha: a << 6 samples
    assert run_case(256, 64, "yuva420p16le", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")      # 16-bit samples: round 5 (strip_hstage_b), the A plane included
    assert not run_case(256, 64, "yuva420p16le", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_u16=1))[0].endswith("+alpha")
    assert not run_case(256, 64, "yuva420p", 192, 48, "yuva420p16le", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")
    assert not run_case(256, 64, "ya8", 192, 48, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0].endswith("+alpha")
    assert not run_case(480, 48, "bgra", 240, 24, "yuva420p", SWS_BICUBIC | BX)[0].endswith("+alpha")                    # narrow: below the planner's width threshold


def test_full_size_batches_and_host_frames():
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    assert run_case(3840, 2160, "bgra", 1920, 1080, "yuva420p", SWS_BICUBIC | BX, seed=2)[0] == "main:rgbread+strip_march+alpha"
    assert run_case(1920, 1080, "yuva420p", 1280, 720, "yuva420p", SWS_LANCZOS | BX, seed=3, device_frames=False)[0] == "main:strip_march+alpha"
    assert run_case(1920, 1080, "yuva444p10le", 1280, 720, "yuva420p", SWS_BICUBIC | SWS_ACCURATE_RND | BX, seed=4)[0] == "main:strip_march+alpha"
    for src, dst, sw, sh, dw, dh, n, flags in (("argb", "yuva420p", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("yuva422p", "yuva444p10le", 1024, 64, 1283, 80, 3, SWS_BICUBIC | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
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
            assert p.path().endswith("+alpha"), p.path()
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)
