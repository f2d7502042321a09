"""-m gpu parity for the 19-bit marching strip kernel (round 5: kernels_stripwide.hpp, sws_k_strip_wide): destinations of 16 bits per component.

Reference arithmetic: hScale8To19_c / hScale16To19_c (swscale.c:69-97, :144-159: int32 lines of 19 bits), yuv2planeX_16_c / yuv2plane1_16_c /
yuv2nv12cX_16_c (output.c:149-217: 32-bit wrap-around sums from (1 << 14) - 0x40000000, 0x8000 + clip_int16(val >> 15)) for yuv4xxp16 / gray16 / p016, and
yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c (output.c:2424-2610) for planar RGB of 16 bits and float32, which the generic writer's X form computes over the
kernel's int32 sum planes (sws_k_sum_writer).  Every case is compared with the oracle; `no_strip_wide = 1` (the element-per-thread kernels of rounds
1 - 4) is compared with the same expectation."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS, SWS_SPLINE,
                           SWS_FULL_CHR_H_INT)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
T0 = {"strip_min_w": 0}
OLD = dict(T0, no_strip_wide=1)

SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "nv12", "nv21", "nv16", "yuv420p10le", "yuv444p12le", "yuv422p9le", "p010le", "p012le", "yuyv422", "uyvy422", "yuvj420p"]
DST_YUV = ["yuv420p16le", "yuv422p16le", "yuv444p16le", "p016le", "p216le", "p416le", "yuv420p16be"]
DST_RGB = ["gbrp16le", "gbrpf32le", "gbrp16be", "gbrpf32be"]
GEOMS = [(644, 70, 322, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (1284, 36, 428, 12, SWS_BILINEAR), (640, 48, 640, 24, SWS_BICUBIC),
         (640, 48, 320, 48, SWS_LANCZOS)]


@pytest.mark.parametrize("form", ["wide", "old"])
@pytest.mark.parametrize("sfmt", SRC)
def test_planar_16_bit_destinations(form, sfmt):
    for dfmt in DST_YUV:
        for (sw, sh, dw, dh, fl) in GEOMS:
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0 if form == "wide" else OLD)
            long_chroma = fl == SWS_LANCZOS and sfmt in ("yuv444p", "yuv444p12le") and "420" in dfmt or fl == SWS_LANCZOS and sfmt in ("yuv444p", "yuv444p12le") and dfmt in ("yuv422p16le", "p016le", "p216le")
            if form == "wide" and sfmt != "yuvj420p" and not long_chroma:     # (full range into limited range: the 19-bit range conversion keeps the old kernels; a 4:1 Lanczos chroma step: beyond the ring)
                assert "strip" in path, (sfmt, dfmt, sw, dw, path)
            if form == "old":
                assert "strip" not in path, (sfmt, dfmt, path)


@pytest.mark.parametrize("form", ["wide", "old"])
@pytest.mark.parametrize("sfmt", ["yuv420p", "nv12", "yuv444p", "yuv420p10le", "p010le", "yuv422p", "yuyv422"])
def test_planar_rgb_16_bit_and_float_destinations(form, sfmt):
    for dfmt in DST_RGB:
        for (sw, sh, dw, dh, fl) in GEOMS:
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0 if form == "wide" else OLD)
            long_chroma = fl == SWS_LANCZOS and sfmt == "yuv444p"
            if form == "wide" and not (dw & 3) and not long_chroma and not (sw == dw and sfmt in ("yuv444p", "yuyv422", "yuv422p")):     # (identity horizontal filters: the single-pass kernels)     # (the sum-writer route takes widths that are multiples of 4)
                assert path.endswith("+fused_gbrp16") and "strip" in path, (sfmt, dfmt, sw, dw, path)


@pytest.mark.parametrize("fl", [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND])
def test_scalers_and_geometries(fl):
    for (sw, sh, dw, dh) in ((1280, 72, 640, 36), (640, 36, 1280, 72), (900, 40, 452, 33), (770, 33, 384, 47), (1920, 30, 1280, 20), (700, 64, 700, 32), (512, 40, 512, 40), (258, 20, 130, 10)):
        for sfmt, dfmt in (("yuv420p", "yuv420p16le"), ("nv12", "p016le"), ("yuv420p10le", "yuv444p16le"), ("yuv420p", "gbrpf32le"), ("nv12", "gbrp16le"), ("gray8", "gray16le"), ("yuv420p", "gray16le"),
                           ("gray10le", "gray16le")):
            run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw, tune=T0)


@pytest.mark.parametrize("sfmt", ["rgb24", "bgra", "gbrp", "rgb565le", "x2rgb10le", "gbrp10le", "rgb48le", "gbrpf32le"])
def test_rgb_sources_into_16_bit_destinations(sfmt):
    for dfmt in ("yuv420p16le", "yuv444p16le", "p016le", "gbrpf32le", "gbrp16le", "rgb48le"):
        for (sw, sh, dw, dh, fl) in ((644, 40, 516, 32, SWS_BICUBIC), (640, 48, 320, 24, SWS_BICUBIC), (320, 24, 640, 48, SWS_BILINEAR)):
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw + len(sfmt), tune=T0)
            if dfmt in ("yuv420p16le", "p016le") and fl == SWS_BICUBIC and sfmt not in ("gbrp10le", "gbrpf32le"):   # (planar RGB beyond 8 bits has no half-width chroma reader: a 4:1 chroma step, long filters)
                assert "rgbread" in path, (sfmt, dfmt, sw, dw, path)


@pytest.mark.parametrize("dfmt", ["rgb48le", "bgr48le", "rgba64le", "bgra64le", "rgb48be"])
def test_packed_rgb_of_16_bits(dfmt):
    """yuv2rgba64_X_c_template / the full-chroma form over the strip kernel's sums (the generic writer's X form); rows in its _1 / _2 forms keep the old kernels"""
    for sfmt in ("yuv420p", "nv12", "yuv420p10le", "yuv444p", "yuv422p"):
        for (sw, sh, dw, dh, fl) in ((644, 70, 324, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (640, 48, 320, 24, SWS_BICUBIC | SWS_FULL_CHR_H_INT),
                                     (640, 48, 640, 96, SWS_BICUBIC), (640, 48, 644, 24, SWS_LANCZOS)):
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0)
            if (sw, dw) in ((400, 332), (320, 640)) and sfmt != "yuv444p":
                assert path.endswith("+sum_writer"), (sfmt, dfmt, sw, dw, path)


def test_generic_writer_over_the_sums_equals_the_vector_epilogue():
    """three forms of the planar RGB writer behind the 19-bit strip kernel: fused into the chroma launch (the default), the vector epilogue over the sums
    (no_wide_epilogue = 2; what RGB sources keep), the generic writer's X form over the sums (no_wide_epilogue = 1)"""
    for dfmt in ("gbrp16le", "gbrpf32le"):
        assert run_case(640, 48, "yuv420p", 320, 24, dfmt, SWS_BICUBIC | BX, tune=T0)[0].endswith("+fused_gbrp16")
        assert run_case(640, 48, "yuv420p", 320, 24, dfmt, SWS_BICUBIC | BX, tune=dict(T0, no_wide_epilogue=2))[0].endswith("+fullchr_gbrp16")
        assert run_case(1284, 50, "nv12", 484, 34, dfmt, SWS_BILINEAR | BX, tune=dict(T0, no_wide_epilogue=2))[0].endswith("+fullchr_gbrp16")
        assert run_case(640, 48, "bgra", 320, 24, dfmt, SWS_BICUBIC | BX, tune=T0)[0].endswith("+fullchr_gbrp16")
        assert run_case(640, 48, "yuv420p", 320, 24, dfmt, SWS_BICUBIC | BX, tune=dict(T0, no_wide_epilogue=1))[0].endswith("+sum_writer")
        assert run_case(644, 50, "nv12", 484, 33, dfmt, SWS_BILINEAR | BX, tune=dict(T0, no_wide_epilogue=1))[0].endswith("+sum_writer")


def test_options_chroma_positions_and_colourspace_details():
    for hp in (-513, 0, 256):
        for vp in (-513, 128):
            opts = dict(dither=1, src_range=0, dst_range=0, src_h_chr_pos=hp, src_v_chr_pos=vp, dst_h_chr_pos=0, dst_v_chr_pos=128, threads=1)
            run_case(1284, 60, "yuv420p", 642, 30, "yuv420p16le", SWS_BICUBIC | BX, seed=abs(hp + vp) + 1, opts=opts, tune=T0)
            run_case(640, 60, "nv12", 320, 30, "gbrpf32le", SWS_BILINEAR | BX, seed=abs(hp + vp) + 2, opts=opts, tune=T0)
    for cs in ((1, 0, 1, 1, 0, 1 << 16, 1 << 16), (5, 1, 9, 0, 1 << 12, 3 << 15, 1 << 15), (9, 0, 9, 0, 0, 1 << 16, 1 << 16)):
        run_case(640, 48, "yuv420p", 320, 24, "gbrpf32le", SWS_BICUBIC | BX, seed=9, colorspace=cs, tune=T0)
        run_case(640, 48, "yuv420p10le", 480, 36, "gbrp16le", SWS_BICUBIC | BX, seed=10, colorspace=cs, tune=T0)


def test_planner_and_fallbacks():
    assert "strip" in run_case(1920, 54, "yuv420p", 1280, 36, "yuv420p16le", SWS_BICUBIC | BX)[0]                     # wide enough without the option
    assert "strip" not in run_case(480, 48, "yuv420p", 240, 24, "yuv420p16le", SWS_BICUBIC | BX)[0]                   # narrow: the old kernels
    assert "strip" in run_case(640, 48, "yuv420p16le", 320, 24, "yuv420p16le", SWS_BICUBIC | BX, tune=T0)[0]          # 16-bit samples: top bit flipped, per-column addend (tests/test_gpu_strip_u16.py)
    assert "strip" not in run_case(640, 48, "yuv420p16le", 320, 24, "yuv420p16le", SWS_BICUBIC | BX, tune=dict(T0, no_strip_u16=1))[0]
    assert "rgbread" in run_case(640, 48, "rgb24", 320, 24, "yuv420p16le", SWS_BICUBIC | BX, tune=T0)[0]             # RGB sources: the reader pre-pass's planes
    assert "strip" not in run_case(640, 48, "yuva420p", 320, 24, "yuva420p16le", SWS_BICUBIC | BX, tune=T0)[0]        # a scaled alpha plane
    assert "strip" not in run_case(1280, 96, "yuv420p", 160, 12, "yuv420p16le", SWS_BICUBIC | BX, tune=T0)[0]         # 8:1: filters beyond the ring
    run_case(642, 48, "yuv420p", 322, 24, "gbrpf32le", SWS_BICUBIC | BX, tune=T0)                                    # width not a multiple of 4
    run_case(640, 48, "yuv420p", 640, 48, "gbrpf32le", SWS_BICUBIC | BX, tune=T0)                                    # same size: the single-pass kernels
    run_case(640, 48, "yuv420p", 320, 24, "gbrapf32le", SWS_BICUBIC | BX, tune=T0)                                   # alpha plane to fill
    # sources of one row: one-tap vertical banks with 4095 / 0 rows -- the semi-planar 16-bit chroma writer has no one-tap form (yuv2nv12cX_16_c multiplies)
    assert "strip" in run_case(1840, 1, "yuv444p", 1799, 55, "p016le", SWS_BILINEAR | BX, tune=T0)[0]
    assert "strip" in run_case(1840, 2, "yuv420p", 1800, 40, "p016le", SWS_BICUBIC | BX, tune=T0)[0]
    assert "strip" in run_case(1840, 1, "yuv444p", 1799, 55, "yuv444p16le", SWS_BILINEAR | BX, tune=T0)[0]


def test_full_size_frames_batches_and_host_frames():
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    assert "strip" in run_case(1920, 1080, "nv12", 640, 360, "gbrpf32le", SWS_BILINEAR | BX, seed=2)[0]               # decoder output -> planar float RGB for inference
    assert "strip" in run_case(3840, 2160, "yuv420p", 1920, 1080, "gbrpf32le", SWS_BICUBIC | BX, seed=3, device_frames=False)[0]
    assert "strip" in run_case(3840, 2160, "yuv420p10le", 1920, 1080, "p016le", SWS_LANCZOS | BX, seed=4)[0]
    assert "strip" in run_case(1920, 1080, "yuv420p", 1280, 720, "yuv444p16le", SWS_BICUBIC | BX, seed=5)[0]
    # a batch equals its frames one by one
    sw, sh, dw, dh, n = 1280, 96, 640, 48, 5
    o = OL.Oracle(sw, sh, "nv12", dw, dh, "gbrpf32le", SWS_BICUBIC | BX)
    p = SwsContext(sw, sh, "nv12", dw, dh, "gbrpf32le", SWS_BICUBIC | BX)
    srcs, dsts, refs = [], [], []
    for k in range(n):
        s = OL.fill_random(OL.Frame("nv12", sw, sh), 60 + k)
        ref = OL.Frame("gbrpf32le", dw, dh, fill=0)
        assert o.scale(s, ref) >= 0
        hs = HostFrame("nv12", sw, sh)
        for a, b in zip(hs.planes, s.planes):
            a[:] = b
        srcs.append(DeviceFrame("nv12", sw, sh).upload(hs)); dsts.append(DeviceFrame("gbrpf32le", dw, dh)); refs.append(ref)
    torch.cuda.synchronize()
    assert p.scale_frames(srcs, dsts) == n
    p.sync()
    assert "strip" in p.path(), p.path()
    for k in range(n):
        out = dsts[k].download()
        for pl, (a, b) in enumerate(zip(out.planes, refs[k].planes)):
            rb = out.row_bytes[pl]
            assert np.array_equal(a[:, :rb], b[:, :rb]), (k, pl)
    p.close()
