"""-m gpu parity for the packed destinations behind the strip kernels (dev_prepare_on: fullchr_on == 4, path suffix +sum_writer): rgb565 / 555 / 444,
x2rgb10 / x2bgr10 (yuv2rgb_X_c_template + yuv2rgb_write, yuv2rgb_full_X_c_template + yuv2rgb_write_full: output.c:1714-1784, :2005-2070), the 8-bit packed
4:4:4 formats (yuv2ayuv_X_c_template, yuv2vyu444_X_c: :2903-3290) and the packed YUV formats of 10 / 12 bits (yuv2y2xxle_X_c, yuv2xv30le / v30xle_X_c,
yuv2xv36le_X_c: :2712-2866, :3088-3169).  The strip kernels leave the vertical sums as int32 planes and the generic writer's X form runs over them with the
bank of the two taps {1, 0} (k_generic_dst.hip sws_k_sum_writer).  Every case also runs with the option no_rgbread_kinds = 2 (the two-pass kernels)."""
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_AREA, SWS_POINT, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT,
                           SWS_FULL_CHR_H_INP)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
TUNE = dict(strip_min_w=0)
OLD = dict(strip_min_w=0, no_rgbread_kinds=2)

SRC = ["yuv420p", "yuv422p", "yuv444p", "nv12", "yuv420p10le", "yuv444p12le", "p010le", "bgra", "rgb24", "gbrp", "x2rgb10le", "rgb565le", "y210le", "yuyv422"]
DST = ["rgb565le", "bgr565le", "rgb555le", "bgr555le", "rgb444le", "bgr444le", "rgb565be", "x2rgb10le", "x2bgr10le", "ayuv", "vuya", "vuyx", "uyva", "vyu444",
       "y210le", "y212le", "xv30le", "v30xle", "xv36le", "xv36be"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    seen = False
    for (sw, sh, dw, dh) in ((256, 64, 192, 48), (320, 50, 512, 80), (132, 34, 68, 18)):
        path, _ = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=TUNE)
        seen = seen or "sum_writer" in path
        old, _ = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=OLD)
        assert "sum_writer" not in old, old
    if src in ("yuv420p", "yuv444p", "nv12", "yuv420p10le", "bgra", "rgb24") and not (src == "bgra" and dst in ("ayuv", "vuya", "uyva")):   # (alpha into alpha: scaled through the luma filters)
        assert seen, (src, dst)


@pytest.mark.parametrize("flags", [SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_LANCZOS, SWS_POINT, SWS_BICUBIC | SWS_ACCURATE_RND, SWS_BICUBIC | SWS_FULL_CHR_H_INT,
                                   SWS_BILINEAR | SWS_FULL_CHR_H_INT | SWS_FULL_CHR_H_INP | SWS_ACCURATE_RND],
                         ids=["area", "bilinear", "bicubic", "lanczos", "point", "accurate", "fullchr", "fullchr_inp"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (640, 96, 428, 64), (260, 200, 520, 300), (1924, 34, 1284, 22), (64, 40, 1032, 44), (2052, 20, 1028, 10), (640, 48, 640, 48),
                                  (640, 48, 640, 96), (640, 96, 320, 96)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for src, dst in (("yuv420p", "rgb565le"), ("nv12", "x2rgb10le"), ("yuv420p10le", "y210le"), ("bgra", "vuya"), ("yuv444p", "xv30le"), ("yuv422p", "bgr444le"), ("rgb24", "xv36le")):
        run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=7, tune=TUNE)


@pytest.mark.parametrize("dither", [0, 1, 2, 3, 4, 5])
def test_dither_modes(dither):
    for dst in ("rgb565le", "bgr555le", "rgb444le", "x2rgb10le"):
        run_case(640, 48, "yuv420p", 320, 24, dst, SWS_BICUBIC | BX, seed=9, tune=TUNE, opts=dict(dither=dither))
        run_case(640, 48, "yuv420p", 320, 24, dst, SWS_BICUBIC | SWS_FULL_CHR_H_INT | BX, seed=9, tune=TUNE, opts=dict(dither=dither))


def test_planner_and_fallbacks():
    assert "sum_writer" in run_case(1920, 54, "yuv420p", 1280, 36, "rgb565le", SWS_BICUBIC | BX)[0]                  # wide enough without the option
    assert "sum_writer" in run_case(1920, 54, "nv12", 1280, 36, "y210le", SWS_BICUBIC | BX)[0]
    assert "sum_writer" in run_case(642, 48, "yuv420p", 322, 24, "rgb565le", SWS_BICUBIC | BX, tune=TUNE)[0]         # a width of 4 k + 2 (round 5)
    assert "sum_writer" not in run_case(642, 48, "yuv420p", 323, 24, "rgb565le", SWS_BICUBIC | BX, tune=TUNE)[0]     # odd width
    assert "sum_writer" in run_case(640, 48, "yuv420p", 320, 24, "rgb48le", SWS_BICUBIC | BX, tune=TUNE)[0]          # 19-bit intermediates: round 5 (sws_k_strip_wide's sums)
    assert "sum_writer" not in run_case(640, 48, "yuv420p", 320, 24, "rgb48le", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_wide=1))[0]
    assert "sum_writer" not in run_case(640, 48, "yuv420p", 320, 24, "y216le", SWS_BICUBIC | BX, tune=TUNE)[0]
    assert "sum_writer" not in run_case(640, 48, "yuva420p", 320, 24, "ayuv", SWS_BICUBIC | BX, tune=TUNE)[0]        # an alpha plane the destination wants scaled
    assert "sum_writer" not in run_case(640, 48, "gray8", 320, 24, "rgb565le", SWS_BICUBIC | BX, tune=TUNE)[0]
    run_case(640, 48, "yuv422p", 320, 48, "rgb565le", SWS_BICUBIC | BX, tune=TUNE)      # one vertical tap each: the writers' _1 forms stay on the generic kernels
    run_case(640, 48, "yuv420p", 320, 48, "rgb565le", SWS_BILINEAR | BX, tune=TUNE)     # two chroma taps adding up to 4096: the _1 / _2 forms


@pytest.mark.parametrize("dst", ["y210le", "y212le", "xv30le", "v30xle", "xv36le", "xv36be"])
@pytest.mark.parametrize("src", ["yuv444p", "yuv422p", "yuv422p10le", "gbrp14le", "bgra", "rgb565le"])
def test_one_tap_vertical_banks_keep_their_coefficients(src, dst):
    """a source of fewer than four rows gives BOTH vertical banks one tap (initFilter, utils.c:575-610), and the packed YUV formats of 10 / 12 bits have X writers
    only (output.c:2712-2866, :3088-3169): the one tap multiplies -- 4095 after the normalisation, 0 in the zero-vector rows a shifted chroma position leaves at
    the top of the picture (chroma 0 there) -- where the other packed writers' _1 forms ignore it.  Found by the round-4 random draw (seed 2222, case 9759)."""
    for (sw, sh, dw, dh, flags, opts) in ((1836, 3, 1512, 39, SWS_BILINEAR | SWS_FULL_CHR_H_INT | SWS_ACCURATE_RND, dict(src_v_chr_pos=256)),
                                          (1836, 2, 1512, 39, SWS_BILINEAR | BX, dict(src_v_chr_pos=256)),
                                          (640, 3, 320, 12, SWS_BILINEAR | BX, dict(src_v_chr_pos=512)),
                                          (640, 3, 320, 40, SWS_LANCZOS | BX, dict(src_v_chr_pos=384)),
                                          (640, 3, 428, 3, SWS_BICUBIC | BX, None),
                                          (640, 2, 320, 7, SWS_BICUBIC | BX, dict(dst_v_chr_pos=256))):
        run_case(sw, sh, src, dw, dh, dst, flags, seed=sw + dh, opts=opts, tune=TUNE)


def test_full_size_frames_host_frames_and_unaligned():
    assert "sum_writer" in run_case(1920, 1080, "yuv420p", 1280, 720, "rgb565le", SWS_BICUBIC | BX, seed=2)[0]
    assert "sum_writer" in run_case(1920, 1080, "nv12", 2560, 1440, "x2rgb10le", SWS_BICUBIC | BX, seed=3)[0]
    assert "sum_writer" in run_case(2560, 1440, "rgb24", 1920, 1080, "vuya", SWS_LANCZOS | BX, seed=4, device_frames=False)[0]
    assert "sum_writer" in run_case(3840, 2160, "yuv420p10le", 1920, 1080, "y210le", SWS_BICUBIC | BX, seed=5)[0]
    from test_gpu_unaligned_frames import run_odd
    for src, dst in (("yuv420p", "rgb565le"), ("nv12", "vyu444"), ("bgra", "x2rgb10le")):
        for pad, shift, flip in ((4, 4, 0), (0, 0, 3), (12, 8, 1)):
            run_odd(640, 40, src, 428, 26, dst, SWS_BICUBIC | BX, pad, shift, flip, tune=TUNE)
