"""-m gpu parity for sws_k_rgbsrc_unity (kernels_rgbsrc.hpp): packed 24 / 32 bpp RGB into 8-bit 4:2:0 / 4:2:2 YUV of the same size.
Readers rgb24ToY_c / rgb24ToUV_half_c (input.c:1068-1172) and the 32-bit rows of rgb16_32To*_c_template (:264-393); the chroma goes
through whatever vertical filter the scaler flag gives the 2:1 chroma step."""
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SINC, SWS_SPLINE, SWS_FAST_BILINEAR, SWS_FULL_CHR_H_INP)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
PATH = "main:rgbsrc_unity"

SRC = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr", "gbrp", "gbrap"]   # (planar 8-bit GBR: the same arithmetic, three planes)
DST = ["yuv420p", "yuv422p", "nv12", "nv21", "nv16", "yuvj420p"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (w, h) in ((256, 64), (322, 50), (129, 33), (67, 18), (1026, 21)):
        path, _ = run_case(w, h, src, w, h, dst, SWS_BICUBIC | BX, seed=w)
        # (a yuvj destination is a range conversion: round 5 gave the wave-march form RNG instantiations -- widths that are multiples of 4; the others
        #  take the strip kernels or the generic ones)
        if dst == "yuvj420p" and (w & 3):
            continue
        if not w & 1:     # (odd widths: the chroma readers are not the "half" forms, chroma is scaled horizontally)
            want = PATH
            if (src, dst) == ("bgr24", "yuv420p"):
                want = "unscaled:bgr24ToYv12"     # the reference's special converter (bgr24ToYv12Wrapper, swscale_unscaled.c:2062-2077)
            assert path == want, (path, w, h)


@pytest.mark.parametrize("flags", [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_SINC,
                                   SWS_BICUBIC | SWS_ACCURATE_RND, SWS_FAST_BILINEAR],
                         ids=["point", "area", "bilinear", "bicubic", "gauss", "lanczos", "spline", "sinc", "accurate", "fast_bilinear"])
@pytest.mark.parametrize("geom", [(640, 96), (130, 200), (4, 2), (2, 2), (1, 1), (5, 3), (1922, 34)], ids=lambda g: f"{g[0]}x{g[1]}")
def test_scalers_and_ragged_sizes(flags, geom):
    w, h = geom
    for src, dst in (("rgb24", "yuv420p"), ("bgra", "nv12"), ("argb", "yuv422p"), ("gbrp", "yuv420p")):
        path, _ = run_case(w, h, src, w, h, dst, flags | BX, seed=7)
        if flags != SWS_FAST_BILINEAR and not (flags in (SWS_SINC, SWS_SPLINE) and dst != "yuv422p") and not w & 1:
            assert path == PATH, path     # (fast bilinear has its own horizontal functions; the 2:1 chroma filters of sinc and spline have more than 16 taps)


def test_the_wave_march_form_and_the_first_form_agree_with_the_oracle():
    """round 4: sws_k_rgbsrc_unity2 (register ring of row pairs, v_dot2 vertical chroma) takes widths that are multiples of 4; the first form keeps the
    others and everything under `no_rgbsrc2` -- both are compared with the oracle on the same cases"""
    from librempeg_amd import SwsContext
    for w, want in ((640, "sws_k_rgbsrc_unity2"), (642, "sws_k_rgbsrc_unity")):
        p = SwsContext(w, 48, "bgra", w, 48, "nv12", SWS_BICUBIC | BX)
        assert p.path() == PATH and p.kernel_name() == want, (w, p.kernel_name())
        p.close()
    for flags in (SWS_POINT, SWS_BILINEAR, SWS_BICUBIC, SWS_LANCZOS, SWS_GAUSS, SWS_AREA):
        for src, dst in (("rgb24", "yuv420p"), ("bgra", "nv12"), ("abgr", "nv21"), ("argb", "yuv422p"), ("gbrp", "nv16"), ("rgb0", "yuv420p")):
            for (w, h) in ((640, 96), (1028, 50), (260, 201), (4, 2), (1924, 34), (512, 7)):
                for tune in ({}, {"no_rgbsrc2": 1}):
                    assert run_case(w, h, src, w, h, dst, flags | BX, seed=w + h, tune=tune)[0] == PATH


def test_other_shapes_keep_their_kernels():
    assert run_case(640, 48, "rgb24", 640, 48, "yuv444p", SWS_BICUBIC | BX)[0] != PATH          # chroma is scaled up horizontally
    assert run_case(640, 48, "rgb24", 640, 48, "yuv420p", SWS_BICUBIC | SWS_FULL_CHR_H_INP | BX)[0] != PATH
    assert run_case(640, 48, "rgb24", 640, 40, "yuv420p", SWS_BICUBIC | BX)[0] != PATH
    assert run_case(640, 48, "rgba", 640, 48, "yuva420p", SWS_BICUBIC | BX)[0] != PATH
    assert run_case(640, 48, "rgb24", 640, 48, "yuv420p10le", SWS_BICUBIC | BX)[0] != PATH
    assert run_case(640, 48, "rgb24", 640, 48, "yuv420p", SWS_BICUBIC | BX, tune=dict(no_rgbsrc=1))[0] == "main:fused_generic_unity"


def test_full_size_frames():
    for src, dst in (("rgb24", "yuv420p"), ("bgra", "nv12")):
        assert run_case(1920, 1080, src, 1920, 1080, dst, SWS_BICUBIC | BX, seed=2)[0] == PATH
    assert run_case(3840, 2160, "rgb24", 3840, 2160, "yuv420p", SWS_BILINEAR | BX, seed=3)[0] == PATH


def test_batches_bands_and_host_frames():
    """several frames per sws_scale_frames() call (grid z), a picture tall enough for many bands per frame (the chroma ring is primed from
    rows above the band), and the host-pointer entry (sws_scale() on pageable memory)."""
    import numpy as np
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    for src, dst, w, h, n, flags in (("rgb24", "yuv420p", 260, 700, 3, SWS_LANCZOS | BX), ("bgra", "nv12", 64, 1030, 5, SWS_BICUBIC | BX),
                                     ("argb", "yuv422p", 516, 90, 4, SWS_BILINEAR | BX)):
        o = OL.Oracle(w, h, src, w, h, dst, flags)
        p = SwsContext(w, h, src, w, h, dst, flags)
        refs, srcs, dsts = [], [], []
        for k in range(n):
            s = OL.fill_random(OL.Frame(src, w, h), 90 + k)
            ref = OL.Frame(dst, w, h)
            assert o.scale(s, ref) == h
            refs.append(ref)
            hs = HostFrame(src, w, h)
            for a, b in zip(hs.planes, s.planes):
                a[:] = b
            srcs.append(DeviceFrame(src, w, h).upload(hs))
            dsts.append(DeviceFrame(dst, w, h))
        torch.cuda.synchronize()
        assert p.scale_frames(srcs, dsts) == n and p.path() == PATH
        p.sync()
        for k in range(n):
            out = dsts[k].download(HostFrame(dst, w, h))
            for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k)
        p.close()
    assert run_case(322, 240, "rgba", 322, 240, "nv21", SWS_BICUBIC | BX, seed=4, device_frames=False)[0] == PATH


PATH444 = "main:rgb_yuv444_unity"


@pytest.mark.parametrize("src", SRC)
def test_rgb_to_yuv444_unity(src):
    """sws_k_rgb_yuv444_unity: 8-bit RGB into planar 8-bit 4:4:4 YUV of the same size (all filters the identity, full chroma readers)"""
    for i, (w, h) in enumerate(((256, 64), (322, 50), (129, 33), (67, 18), (1026, 21), (1, 1), (3, 2), (1920, 1080))):
        for flags in (SWS_BICUBIC, SWS_POINT | SWS_ACCURATE_RND):
            path, _ = run_case(w, h, src, w, h, "yuv444p", flags | BX, seed=w + i)
            assert path == PATH444, (path, w, h)
        run_case(w, h, src, w, h, "yuv444p", SWS_BILINEAR | BX, seed=i, device_frames=False)
    assert run_case(640, 48, src, 640, 48, "yuvj444p", SWS_BICUBIC | BX)[0] != PATH444        # a range conversion
    assert run_case(640, 48, src, 640, 48, "yuv444p10le", SWS_BICUBIC | BX)[0] != PATH444
    assert run_case(640, 48, src, 640, 48, "yuv444p", SWS_FAST_BILINEAR | BX)[0] in (PATH444, "main:fused_generic_unity", "main:two_pass", "main:fused_tile", "main:rgbread+strip_march")
