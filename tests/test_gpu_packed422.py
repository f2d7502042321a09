"""-m gpu parity for packed 4:2:2 destinations through the scaler (dev_exec.hip: planar writers + the streaming interleave, "+join422"):
yuv2422_X_c_template / yuv2422_1 with one chroma tap (output.c:843-1000) against yuv2planeX_8_c / yuv2plane1_8_c (:438-493) on 8-bit sources;
the short vertical forms (vscale.c:136-158) keep the packed writer of the generic kernels."""
import numpy as np
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_FAST_BILINEAR)
AR = SWS_ACCURATE_RND
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
TUNE = dict(strip_min_w=0)

SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "nv12", "nv21", "rgb24", "bgra", "yuvj420p", "gbrp"]
DST = ["yuyv422", "uyvy422", "yvyu422"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh, fl) in ((256, 64, 256, 64, SWS_BICUBIC), (256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 66, 17, SWS_AREA),
                                 (256, 64, 320, 96, SWS_BILINEAR), (256, 64, 256, 64, SWS_BILINEAR), (256, 64, 250, 64, SWS_LANCZOS), (130, 30, 131, 31, SWS_BICUBIC)):
        for tune in (None, TUNE):
            r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=tune)
            if r and fl == SWS_BICUBIC and not dw & 1 and (src, dst) not in (("yuv422p", "yuyv422"), ("yuv422p", "uyvy422")) and not (src in ("yuv422p",) and (sw, sh) == (dw, dh)):
                if not ((sw, sh) == (dw, dh) and src in ("yuv422p", "yuv444p", "rgb24", "bgra", "gbrp")):      # (a one- or two-tap chroma filter may be a short form)
                    assert r[0].endswith("join422"), (r[0], src, dst, sw, sh, dw, dh)


def test_short_vertical_forms_keep_the_packed_writer():
    # bilinear 2x vertical upscale: two luma and two chroma taps (yuv2422_2); same-size 4:2:0 -> 4:2:2 bilinear: one luma, two chroma taps (yuv2422_1)
    assert not run_case(256, 64, "yuv420p", 256, 128, "yuyv422", SWS_BILINEAR | BX)[0].endswith("+join422")
    # (round 5: ... whose blend -- the first chroma row below 2048, the mean of both rows from there on -- is the X arithmetic over substituted taps: the planar writers + join, or the one-pass kernel)
    assert run_case(640, 64, "yuv420p", 640, 64, "yuyv422", SWS_BILINEAR | BX)[0] == "main:mixed_join422"
    assert run_case(640, 64, "yuv420p", 320, 64, "yuyv422", SWS_BILINEAR | BX)[0].endswith("+join422")
    assert not run_case(640, 64, "yuv420p", 640, 64, "yuyv422", SWS_BILINEAR | BX, tune=dict(no_short_forms=1))[0].endswith("join422")
    assert run_case(256, 64, "yuv420p10le", 256, 64, "yuyv422", SWS_BICUBIC | BX)[0].endswith("+join422")     # (the ordered dither belongs to the planar 8-bit writers only)
    assert not run_case(255, 64, "yuv420p", 255, 64, "yuyv422", SWS_BICUBIC | BX)[0].endswith("+join422")         # odd width: the last pair
    assert run_case(640, 64, "yuv420p", 640, 64, "yuyv422", SWS_BICUBIC | BX)[0] == "main:mixed_join422"       # (round 5: plane pass, chroma strip launch and interleave as one pass)
    assert run_case(640, 64, "yuv420p", 640, 64, "yuyv422", SWS_BICUBIC | BX, tune=dict(no_wave=1))[0] == "main:plane1+strip_chroma+join422"
    assert run_case(644, 64, "yuv420p", 644, 64, "yuyv422", SWS_BICUBIC | BX)[0] == "main:plane1+strip_chroma+join422"      # (whole groups of 8 pixels only)
    assert run_case(256, 64, "yuv420p", 256, 64, "yuyv422", SWS_BICUBIC | BX)[0] == "main:fused_generic_unity+join422"      # (narrow pictures: no mixed plan)
    assert run_case(256, 64, "yuv420p", 256, 64, "uyvy422", SWS_BICUBIC | BX, tune=dict(no_mixed=1))[0] == "main:fused_generic_unity"


@pytest.mark.parametrize("src", ["yuv420p", "nv12", "nv21", "yuv410p", "yuv440p", "yuvj420p", "yuv411p"])
@pytest.mark.parametrize("dst", DST)
def test_same_size_one_pass(src, dst):
    """sws_k_mixed_join422: vertical chroma filters of every scaler, odd heights, one-row pictures, host frames, sources with other horizontal chroma steps (not its shape)"""
    from librempeg_amd import SWS_GAUSS, SWS_SPLINE, SWS_SINC, SWS_POINT
    for (w, h) in ((256, 64), (1920, 1080), (640, 37), (64, 1), (32, 2), (3840, 6)):
        for fl in (SWS_BICUBIC, SWS_LANCZOS, SWS_AREA, SWS_GAUSS, SWS_SPLINE, SWS_SINC, SWS_POINT, SWS_BICUBIC | AR, SWS_BILINEAR, SWS_FAST_BILINEAR):
            if (w, h) == (1920, 1080) and fl not in (SWS_BICUBIC, SWS_LANCZOS, SWS_BILINEAR):
                continue
            r = run_case(w, h, src, w, h, dst, fl | BX, seed=w + h, tune=TUNE)
            if fl in (SWS_BICUBIC, SWS_BILINEAR) and src in ("yuv420p", "nv12", "nv21") and h > 2:
                assert r[0] == "main:mixed_join422", (r[0], src, dst, w, h)
    run_case(1280, 720, src, 1280, 720, dst, SWS_BICUBIC | BX, seed=5, device_frames=False)
    opts = dict(dither=1, src_range=0, dst_range=0, src_h_chr_pos=-513, src_v_chr_pos=128, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
    run_case(640, 48, src, 640, 48, dst, SWS_BICUBIC | BX, seed=6, opts=opts)


def test_full_size_batches_and_host_frames():
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    assert run_case(1920, 1080, "yuv420p", 1920, 1080, "yuyv422", SWS_BICUBIC | BX, seed=2)[0] == "main:mixed_join422"
    assert run_case(3840, 2160, "nv12", 3840, 2160, "uyvy422", SWS_BICUBIC | BX, seed=7)[0] == "main:mixed_join422"
    assert run_case(1920, 1080, "yuv420p", 1280, 720, "uyvy422", SWS_BICUBIC | BX, seed=3, device_frames=False)[0].endswith("+join422")
    assert run_case(1920, 1080, "bgra", 1280, 720, "yuyv422", SWS_BICUBIC | BX, seed=4)[0].endswith("+join422")
    for src, dst, sw, sh, dw, dh, n, flags in (("yuv420p", "yuyv422", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("nv12", "uyvy422", 1024, 64, 1024, 64, 3, SWS_BICUBIC | BX)):
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
            assert p.path().endswith("join422"), p.path()
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)


PSRC = ["yuyv422", "uyvy422", "yvyu422"]


@pytest.mark.parametrize("src", PSRC)
@pytest.mark.parametrize("dst", ["yuv420p", "yuv422p", "nv12", "yuv420p10le", "rgb24", "bgra", "yuyv422", "uyvy422", "yuv444p"])
def test_packed_sources(src, dst):
    """packed 4:2:2 sources of the scaler: the streaming de-interleave into a planar working picture, then the kernels of a planar 8-bit
    source ("main:split422+...")"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 66, 17, SWS_AREA), (256, 64, 320, 96, SWS_BILINEAR),
                                 (256, 64, 250, 64, SWS_LANCZOS), (130, 30, 131, 31, SWS_BICUBIC), (256, 64, 128, 32, SWS_FAST_BILINEAR), (256, 64, 128, 32, SWS_POINT)):
        # (half-width-chroma YUV destinations at ratios the plain strip plan takes: the lockstep strip kernel reads the packed frame itself, "main:strip_packed422";
        #  no_strip_rgbsrc keeps the split pass in the test)
        for tune in (None, TUNE, dict(TUNE, no_strip_rgbsrc=1)):
            r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=tune)
            assert r[0].startswith("main:split422+") or r[0].startswith("main:strip_packed422"), (r[0], src, dst, sw, dw)
            if tune and "no_strip_rgbsrc" in tune:
                assert r[0].startswith("main:split422+"), (r[0], src, dst, sw, dw)
            elif tune and dst in ("yuv420p", "yuv422p", "nv12", "yuv420p10le", "yuyv422", "uyvy422") and (sw, dw) == (256, 192):
                assert r[0].startswith("main:strip_packed422"), (r[0], src, dst, sw, dw)
    assert not run_case(255, 64, src, 128, 32, dst, SWS_BICUBIC | BX)[0].startswith("main:split422+")      # odd source width: the readers keep it


def test_packed_sources_full_size_and_batches():
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    assert run_case(1920, 1080, "yuyv422", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2)[0] == "main:strip_packed422"
    assert run_case(1920, 1080, "yuyv422", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2, tune=dict(no_strip_rgbsrc=1))[0] == "main:split422+strip_march"
    assert run_case(3840, 2160, "uyvy422", 1920, 1080, "nv12", SWS_BILINEAR | BX, seed=5)[0] == "main:strip_packed422"
    assert run_case(1920, 1080, "yvyu422", 1280, 720, "p010le", SWS_LANCZOS | BX, seed=6)[0] == "main:strip_packed422"
    assert run_case(1920, 1080, "uyvy422", 1280, 720, "nv12", SWS_BILINEAR | BX, seed=3, device_frames=False)[0] == "main:strip_packed422"
    assert run_case(1920, 1080, "yuyv422", 1280, 720, "rgb24", SWS_BICUBIC | BX, seed=4)[0] == "main:split422+strip_rgb"
    for src, dst, sw, sh, dw, dh, n, flags in (("yuyv422", "yuv420p", 1284, 70, 1028, 56, 5, SWS_BICUBIC | BX), ("uyvy422", "yuyv422", 1024, 64, 1280, 80, 3, SWS_BICUBIC | BX)):
        o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
        p = SwsContext(sw, sh, src, dw, dh, dst, flags)
        refs, srcs, dsts = [], [], []
        for k in range(n):
            s = OL.fill_random(OL.Frame(src, sw, sh), 80 + k)
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
            assert p.path().startswith("main:split422+") or p.path().startswith("main:strip_packed422"), p.path()
            for k in range(n):
                out = dsts[k].download()
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, k, rep)


@pytest.mark.parametrize("src", ["nv12", "nv21", "nv16", "nv24", "nv42"])
@pytest.mark.parametrize("dst", ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0"])
def test_semi_planar_sources_into_rgb(src, dst):
    """decoder output scaled for display: the interleaved chroma plane is split into planar working planes (nvXXtoUV_c copies bytes), then the
    strip kernel with the RGB epilogue as for a planar source ("main:splitnv+strip_rgb"); round 4: on aligned frames the strip-RGB kernel reads the
    interleaved plane itself where its LDS-DMA form applies ("main:nvdirect+strip_rgb", tests/test_gpu_strip_short.py)"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 66, 18, SWS_AREA), (256, 64, 320, 96, SWS_BILINEAR),
                                 (256, 64, 250, 64, SWS_LANCZOS), (130, 30, 131, 31, SWS_BICUBIC), (131, 31, 200, 40, SWS_BICUBIC | SWS_ACCURATE_RND)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh)
        # (an odd destination width and a 4:4:4 source force the full-chroma writers, utils.c:1330-1349: not the LUT writers this path is for)
        if not dw & 1 and src not in ("nv24", "nv42"):
            assert r[0].startswith(("main:splitnv+", "main:nvdirect+")), (r[0], src, dst, sw, dw)
    if src not in ("nv24", "nv42"):
        assert run_case(1920, 1080, src, 1280, 720, dst, SWS_BICUBIC | BX, seed=7)[0] == "main:nvdirect+strip_rgb"
    assert not run_case(256, 64, src, 256, 64, dst, SWS_BICUBIC | BX, seed=8)[0].startswith("main:splitnv+")      # same size: sws_k_rgb_march reads nv12 itself


def test_nv12_to_rgb_full_size_is_the_strip_kernel():
    assert run_case(3840, 2160, "nv12", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=2)[0] == "main:nvdirect+strip_rgb"
    assert run_case(1920, 1080, "nv12", 1280, 720, "rgb24", SWS_BILINEAR | BX, seed=3, device_frames=False)[0] == "main:nvdirect+strip_rgb"


@pytest.mark.parametrize("src", ["p010le", "p012le", "p210le", "p010be"])
@pytest.mark.parametrize("dst", ["rgb24", "bgra", "argb", "rgb0"])
def test_p01x_sources_into_rgb(src, dst):
    """10 / 12-bit decoder output scaled for display: both planes become a planar working picture with the samples shifted down
    (p010LEToY_c / p010LEToUV_c, input.c:950-1008), then the 16-bit instantiation of the strip kernel with the RGB epilogue"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_BICUBIC), (132, 34, 66, 18, SWS_AREA), (256, 64, 320, 96, SWS_BILINEAR),
                                 (256, 64, 250, 64, SWS_LANCZOS), (130, 30, 131, 31, SWS_BICUBIC)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh)
        if not dw & 1:
            assert r[0].startswith(("main:splitnv+", "main:nvdirect+")), (r[0], src, dst, sw, dw)
    assert run_case(1920, 1080, src, 1280, 720, dst, SWS_BICUBIC | BX, seed=7)[0] == "main:nvdirect+strip_rgb"
    assert run_case(256, 64, src, 256, 64, dst, SWS_BICUBIC | BX, seed=8)[0].startswith(("main:splitnv+", "main:nvdirect+"))      # same size too: one-tap horizontal banks


def test_same_size_10bit_pictures_and_packed_rgb():
    """identity horizontal filters: a 10 / 12-bit 4:2:x picture into packed RGB (decoded HDR for display) takes the 16-bit strip kernel with the RGB
    epilogue, packed RGB into a 10-bit 4:2:0 picture the reader pre-pass + strip kernel (the 8-bit twins have kernels of their own)"""
    for src in ("yuv420p10le", "yuv422p10le", "yuv420p12le", "yuv420p9le"):
        for dst in ("rgb24", "bgra", "argb", "bgr24"):
            for (w, h, fl) in ((256, 64, SWS_BICUBIC), (322, 50, SWS_LANCZOS), (130, 34, SWS_BICUBIC | SWS_ACCURATE_RND), (64, 18, SWS_BILINEAR), (131, 33, SWS_BICUBIC)):
                r = run_case(w, h, src, w, h, dst, fl | BX, seed=w)
                # (a 4:2:2 source has one luma and one chroma tap here: the packed writers' one-tap form, which rounds differently from the X form of this kernel)
                if fl & SWS_BICUBIC and not w & 1 and "420" in src:
                    assert r[0] == "main:strip_rgb", (r[0], src, dst, w)
    for src in ("rgb24", "bgra", "argb", "gbrp"):
        for dst in ("yuv420p10le", "yuv422p10le", "p010le", "yuv420p12le"):
            for (w, h, fl) in ((256, 64, SWS_BICUBIC), (324, 50, SWS_LANCZOS), (132, 34, SWS_BICUBIC | SWS_ACCURATE_RND), (64, 18, SWS_BILINEAR), (130, 33, SWS_BICUBIC)):
                r = run_case(w, h, src, w, h, dst, fl | BX, seed=w, tune=TUNE)
                if not w & 3 and "422" not in dst:      # (4:2:2: every filter is the identity, the generic one-pass kernel keeps it)
                    assert r[0] in ("main:rgbread+strip_march", "main:strip_rgbsrc"), (r[0], src, dst, w)
    assert run_case(1920, 1080, "yuv420p10le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=3)[0] == "main:strip_rgb"
    assert run_case(1920, 1080, "bgra", 1920, 1080, "yuv420p10le", SWS_BICUBIC | BX, seed=4)[0] == "main:strip_rgbsrc"


def test_one_tap_vertical_forms_through_the_lut_writers():
    """one vertical tap for luma and chroma (4:2:2 / 4:4:0-free sources at an unscaled height): yuv2rgb_1_c_template never looks at the coefficient, i.e. the X arithmetic
    with the tap 4096 -- the strip kernel with the RGB epilogue takes same-size 10-bit 4:2:2 pictures and width-only scaling"""
    for src in ("yuv422p", "yuv422p10le", "yuv422p12le", "nv16", "p210le", "yuyv422", "uyvy422", "yuvj422p", "yuva422p"):
        for dst in ("bgra", "rgb24", "argb", "bgr24", "rgb0"):
            for (sw, sh, dw, dh, fl) in ((256, 64, 256, 64, SWS_BICUBIC), (320, 50, 200, 50, SWS_BICUBIC), (200, 37, 320, 37, SWS_LANCZOS), (1920, 24, 1920, 24, SWS_BILINEAR),
                                         (1920, 24, 480, 24, SWS_BICUBIC), (1920, 24, 1280, 24, SWS_BICUBIC)):
                run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=dict(strip_min_w=0))
    assert run_case(1920, 1080, "yuv422p10le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=51)[0] == "main:strip_rgb"
    assert run_case(1920, 1080, "yuv422p", 1280, 1080, "rgb24", SWS_BICUBIC | BX, seed=52)[0] == "main:strip_rgb"
    assert not run_case(1920, 64, "yuva422p", 1280, 64, "bgra", SWS_BICUBIC | BX, seed=53)[0].endswith("+alpha")      # yuv2rgb_1's alpha: (a * 255 + 16384) >> 15


@pytest.mark.parametrize("src", ["yuv420p10le", "yuv422p10le", "yuv444p12le", "yuv420p9le", "yuv420p16le", "p010le", "yuv422p14le"])
@pytest.mark.parametrize("dst", ["yuyv422", "uyvy422", "yvyu422"])
def test_high_bit_depth_sources_into_packed422(src, dst):
    """9 .. 16-bit sources into the 8-bit packed 4:2:2 formats (SDI-style output of a 10-bit pipeline): the packed writers do not dither (the ordered
    dither of swscale.c:292-300 goes to yuv2planeX_8_c only), so the planar working picture is written without it"""
    for (sw, sh, dw, dh, fl) in ((256, 64, 256, 64, SWS_BICUBIC), (256, 64, 192, 48, SWS_BICUBIC), (320, 50, 512, 80, SWS_LANCZOS), (132, 34, 66, 17, SWS_AREA),
                                 (1920, 32, 1920, 32, SWS_BICUBIC), (1920, 32, 1280, 24, SWS_BICUBIC), (130, 30, 132, 31, SWS_BICUBIC | SWS_ACCURATE_RND), (256, 64, 256, 64, SWS_BILINEAR)):
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=dict(strip_min_w=0))
        if (sw, sh) == (1920, 32) and src not in ("yuv420p16le", "yuv422p10le", "yuv444p12le", "yuv422p14le"):     # (4:2:2 / 4:4:4 sources at an unscaled height: one chroma tap, fine; kept out of the assertion for the two-tap forms)
            assert r[0].endswith("+join422"), (r[0], src, dst, sw, dw)
