"""-m gpu parity for the packed writers' short vertical forms behind the strip kernel with the RGB epilogue (round 5).  packed_vscale (vscale.c:135-157) picks per
output row: yuv2rgb_1_c_template with a chroma blend (one luma tap, two chroma taps that sum to 4096: (u0 (4096 - a) + u1 a + (128 << 11)) >> 19, output.c:1913-1937
-- the X arithmetic on the bank's taps) and yuv2rgb_2_c_template (two taps each that sum to 4096 -- bilinear up-scaling: no rounding constant, output.c:1853-1895).
The planner writes both into the strip plan: the luma tap 4096, a per-row rounding offset.  Compared with the oracle byte for byte, with the
plan (main:strip_rgb) and without it (no_short_forms: the element-per-thread writers)."""
import pytest

from librempeg_amd import SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_BICUBIC, SWS_POINT, SWS_AREA, SWS_BITEXACT, SWS_ACCURATE_RND
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
T0 = dict(strip_min_w=0)
PATHS = ("main:strip_rgb", "main:nvdirect+strip_rgb", "main:split422+strip_rgb", "main:splitnv+strip_rgb")

SRC = ["yuv420p", "yuv422p", "yuv444p", "nv12", "nv21", "yuv410p", "yuvj420p", "yuv420p10le", "yuv422p12le", "yuyv422", "p010le", "yuv440p"]
DST = ["bgra", "rgb24", "argb", "bgr24", "rgb0", "abgr"]
#        two taps each (up, bilinear)              one luma tap + two chroma taps (same height)     mixed ratios
GEOM = [(640, 48, 1280, 96), (640, 48, 642, 97), (640, 48, 320, 48), (640, 48, 640, 48), (322, 31, 644, 31), (640, 24, 1280, 25), (640, 5, 640, 64), (400, 66, 332, 132)]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for k, (sw, sh, dw, dh) in enumerate(GEOM):
        for fl in (SWS_BILINEAR, SWS_FAST_BILINEAR):
            if fl == SWS_FAST_BILINEAR and (SRC.index(src) + DST.index(dst) + k) % 2:
                continue
            r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=T0)
            if k == 0 and fl == SWS_BILINEAR and src in ("yuv420p", "yuv422p", "nv12", "nv21", "yuv420p10le"):      # (yuv444p: full chroma is forced, another route)
                assert r[0] in PATHS, (r[0], src, dst)
            if k < 3:
                old = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=dict(T0, no_short_forms=1))
                if k == 0 and fl == SWS_BILINEAR:
                    assert old[0] not in PATHS, old[0]


@pytest.mark.parametrize("flags", [SWS_BILINEAR, SWS_BILINEAR | SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_BICUBIC, SWS_FAST_BILINEAR | SWS_ACCURATE_RND],
                         ids=["bilinear", "accurate", "point", "area", "bicubic", "fast_accurate"])
def test_scalers(flags):
    """point and area up-scaling have one or two vertical taps too (not always summing to 4096 in two non-negative taps: the X form row by row)"""
    for src, dst in (("yuv420p", "bgra"), ("nv12", "rgb24"), ("yuv422p", "argb"), ("yuv444p", "bgr24"), ("yuv420p10le", "rgb0")):
        for (sw, sh, dw, dh) in GEOM + [(640, 360, 656, 372), (640, 100, 640, 101), (640, 100, 640, 199), (320, 9, 960, 27)]:
            run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=dh, tune=T0)


def test_options_and_full_size():
    opts = dict(dither=1, src_range=1, dst_range=0, src_h_chr_pos=0, src_v_chr_pos=128, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
    for (sw, sh, dw, dh) in GEOM:
        run_case(sw, sh, "yuv420p", dw, dh, "bgra", SWS_BILINEAR | BX, seed=3, opts=opts, tune=T0)
        run_case(sw, sh, "yuva420p", dw, dh, "bgra", SWS_BILINEAR | BX, seed=4, tune=T0)        # (an alpha plane: the writers' own alpha formulas, not this plan)
    assert run_case(1280, 720, "yuv420p", 1920, 1080, "bgra", SWS_BILINEAR, seed=5)[0] == "main:strip_rgb"
    assert run_case(1280, 720, "nv12", 1920, 1080, "rgb24", SWS_BILINEAR | BX, seed=6)[0] in PATHS
    assert run_case(1920, 1080, "yuv420p", 3840, 2160, "bgra", SWS_FAST_BILINEAR, seed=7)[0] == "main:strip_rgb"
    run_case(1920, 1080, "yuv420p", 1280, 1080, "rgb24", SWS_BILINEAR | BX, seed=8, device_frames=False)
    run_case(1280, 720, "yuv420p10le", 1920, 1080, "bgra", SWS_BILINEAR | BX, seed=9)


FULL = [("bgra", "bgra"), ("rgb24", "bgra"), ("bgra", "rgb24"), ("rgba", "argb"), ("bgr24", "rgb24"), ("yuv444p", "bgra"), ("yuv444p10le", "rgb24"), ("gray8", "bgra"), ("gray16le", "rgb24"),
        ("gbrp", "bgra"), ("rgb565le", "bgra"), ("x2rgb10le", "rgb24"), ("yuva444p", "bgra"), ("rgb48le", "bgra"), ("yuvj444p", "abgr"), ("bgra", "gbrp"), ("yuv444p", "gbrap")]


@pytest.mark.parametrize("pair", FULL, ids=lambda c: f"{c[0]}-{c[1]}")
def test_full_chroma_writers(pair):
    """the full-chroma writers' short forms behind the sum planes (sws_k_fullchr_rgb leaves the rounding constant out of the rows of yuv2rgb_full_2_c_template and of
    yuv2rgb_full_1_c_template's chroma blend, output.c:2225-2306): image up-scaling RGB -> RGB with SWS_BILINEAR / SWS_FAST_BILINEAR, 4:4:4 and gray sources; planar RGB
    destinations have the X form only (any_vscale)"""
    from librempeg_amd import SWS_FULL_CHR_H_INT
    src, dst = pair
    for k, (sw, sh, dw, dh) in enumerate(GEOM):
        for fl in (SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_BILINEAR | SWS_FULL_CHR_H_INT):
            r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=T0)
            if k == 0 and fl == SWS_BILINEAR and dst in ("bgra", "rgb24", "argb", "abgr") and src not in ("yuva444p",):
                assert r[0].endswith("+fullchr_rgb") or (r[0] == "main:strip_rgb2rgb" and src in ("bgra", "rgb24", "rgba", "bgr24")), (r[0], src, dst)
                if src in ("bgra", "rgb24", "rgba", "bgr24"):      # (8-bit packed RGB both ways: the one-launch kernel carries the decision in its row entries)
                    assert r[0] == "main:strip_rgb2rgb", r[0]
                    assert run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh, tune=dict(T0, no_strip_rgb2rgb=1))[0].endswith("+fullchr_rgb")
        if k < 3:
            run_case(sw, sh, src, dw, dh, dst, SWS_BILINEAR | BX, seed=sw + dh, tune=dict(T0, no_short_forms=1))
    run_case(1280, 720, src, 1920, 1080, dst, SWS_BILINEAR | BX, seed=11)


def test_full_chroma_from_half_width_chroma_sources():
    from librempeg_amd import SWS_FULL_CHR_H_INT
    for src in ("yuv420p", "nv12", "yuv422p", "yuv420p10le", "yuyv422"):
        for dst in ("bgra", "rgb24"):
            for (sw, sh, dw, dh) in GEOM:
                run_case(sw, sh, src, dw, dh, dst, SWS_BILINEAR | SWS_FULL_CHR_H_INT | BX, seed=dh, tune=T0)
                run_case(sw, sh, src, dw, dh, dst, SWS_FAST_BILINEAR | SWS_FULL_CHR_H_INT | SWS_ACCURATE_RND, seed=dh + 1, tune=T0)


@pytest.mark.parametrize("src", ["rgb24", "bgr24", "rgb0", "bgra", "gbrp", "rgb565le", "x2rgb10le", "gbrp10le", "rgb48le", "gbrpf32le"])
def test_rgb_sources_under_the_lut_writers(src):
    """RGB -> RGB without forced full chroma (SWS_FAST_BILINEAR switches it off, utils.c:1277-1285; so does an ordered dither): the LUT writers over a packed source -- reader
    pre-pass, strip launches on half-width chroma, sws_k_lut_rgb.  Without an alpha plane (bgra -> bgra keeps the element-per-thread writers); all four geometries, bitexact and not"""
    for dst in ("rgb24", "bgra", "bgr24", "argb", "rgb0"):
        for (sw, sh, dw, dh) in ((644, 70, 324, 35), (400, 66, 332, 54), (320, 40, 640, 80), (1366, 36, 1282, 34), (640, 48, 640, 24)):
            for fl in (SWS_FAST_BILINEAR, SWS_FAST_BILINEAR | BX):
                r = run_case(sw, sh, src, dw, dh, dst, fl, seed=sw + dh, tune=T0)
                if (sw, dw) == (400, 332) and src in ("rgb24", "bgr24", "gbrp") and fl == SWS_FAST_BILINEAR | BX:      # (rgb0 into a destination with alpha: its X byte feeds the alpha channel as 255 -- an alpha plane, not this route)
                    assert r[0].endswith("+lut_rgb"), (r[0], src, dst)
            opts = dict(dither=1, src_range=0, dst_range=0, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
            run_case(sw, sh, src, dw, dh, dst, SWS_BILINEAR | BX, seed=sw, opts=opts, tune=T0)
    run_case(1920, 1080, src, 1280, 720, "rgb24", SWS_FAST_BILINEAR, seed=3)


def test_same_size_bilinear_takes_the_march_kernel():
    """4:2:0 -> 24 / 32 bpp RGB at the same size with SWS_BILINEAR (a decoder's nv12 for display): one luma tap, two chroma taps -- yuv2rgb_1_c_template with its blend, the X
    arithmetic on the bank's taps -- on sws_k_rgb_march like the bicubic twin (0.032 -> 0.010 ms per 4K frame); with an alpha plane and under no_short_forms the older kernels"""
    from librempeg_amd import SwsContext
    for src in ("yuv420p", "nv12", "nv21", "yuvj420p", "yuv422p", "yuv410p"):
        for dst in ("bgra", "rgb24", "bgr0", "argb", "bgr24"):
            for (w, h) in ((640, 48), (1920, 1080), (1366, 50), (644, 37), (64, 2)):
                for fl in (SWS_BILINEAR, SWS_BILINEAR | BX, SWS_FAST_BILINEAR | BX, SWS_BILINEAR | SWS_ACCURATE_RND):
                    run_case(w, h, src, w, h, dst, fl, seed=w + h)
            run_case(1280, 720, src, 1280, 720, dst, SWS_BILINEAR | BX, seed=5, device_frames=False)
            run_case(640, 48, src, 640, 48, dst, SWS_BILINEAR | BX, seed=6, tune=dict(no_short_forms=1))
    p = SwsContext(1920, 1080, "nv12", 1920, 1080, "bgra", SWS_BILINEAR | BX)
    assert (p.path(), p.kernel_name()) == ("main:fused_rgb_unity", "sws_k_rgb_march"), (p.path(), p.kernel_name())
    p.close()
    run_case(640, 48, "yuva420p", 640, 48, "bgra", SWS_BILINEAR | BX, seed=7)
