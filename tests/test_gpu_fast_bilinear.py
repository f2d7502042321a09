"""-m gpu parity for SWS_FAST_BILINEAR behind the strip kernels (round 5): ff_hyscale_fast_c / ff_hcscale_fast_c (hscale_fast_bilinear.c:23-55) restated
as two-tap banks of hScale8To15_c (dev_prepare_on: fast_banks) -- luma and alpha {(128 - xalpha) << 7, xalpha << 7}, chroma {(xalpha ^ 127) << 7, xalpha << 7},
the columns at and behind the last source sample 128 x that sample -- so that every plan of the filter path applies to the flag players and capture tools pass
most often.  The oracle keeps the reference's own loops; every case is compared byte for byte, with the banks (the strip kernels, the tile kernels, whatever the
planner picks) and without them (no_fast_banks: the fast functions in the element-per-thread readers)."""
import pytest

from librempeg_amd import SWS_FAST_BILINEAR, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT, SWS_FULL_CHR_H_INP
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
FB, BX = SWS_FAST_BILINEAR, SWS_BITEXACT
T0 = dict(strip_min_w=0)
OLD = dict(no_fast_banks=1)

SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "nv12", "nv21", "yuyv422", "uyvy422", "gray8", "yuva420p", "yuvj420p", "yuv440p", "nv16", "yuv411p"]
DST = ["yuv420p", "nv12", "yuv422p", "yuv444p", "yuv420p10le", "p010le", "bgra", "rgb24", "rgb565le", "yuyv422", "gray8", "yuva420p", "gbrp", "yuvj420p", "argb", "yuv420p16le"]
GEOM = [(644, 70, 324, 35), (400, 66, 332, 54), (320, 40, 640, 80), (640, 48, 640, 48), (640, 48, 640, 24), (1284, 36, 428, 12), (322, 31, 645, 17), (64, 40, 1030, 44),
        (2052, 20, 258, 10), (642, 30, 321, 30), (2, 8, 640, 8), (3, 9, 321, 18), (640, 16, 2, 4), (640, 3, 320, 24)]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for k, (sw, sh, dw, dh) in enumerate(GEOM):
        if k >= 6 and (SRC.index(src) + DST.index(dst) + k) % 3:       # (the odd shapes on a third of the pairs each)
            continue
        r = run_case(sw, sh, src, dw, dh, dst, FB | BX, seed=sw + dh, tune=T0)
        known = src in ("yuv420p", "nv12", "yuv422p", "yuv444p") and dst in ("yuv420p", "nv12", "yuv422p", "yuv444p", "yuv420p10le", "p010le")
        if r and k == 1 and known:
            assert r[0] == "main:strip_march", (r[0], src, dst)
        if k < 4:
            old = run_case(sw, sh, src, dw, dh, dst, FB | BX, seed=sw + dh, tune=dict(T0, **OLD))
            if old and k == 1 and known:
                assert "strip" not in old[0], old[0]


@pytest.mark.parametrize("flags", [FB, FB | BX, FB | SWS_ACCURATE_RND, FB | SWS_FULL_CHR_H_INT, FB | SWS_FULL_CHR_H_INP, FB | SWS_FULL_CHR_H_INT | SWS_ACCURATE_RND | BX],
                         ids=["plain", "bitexact", "accurate", "fullint", "fullinp", "fullint_accurate"])
def test_flag_sets(flags):
    for src, dst in (("yuv420p", "bgra"), ("yuv420p", "rgb24"), ("nv12", "bgr24"), ("yuv444p", "gbrp"), ("yuv420p", "yuv444p"), ("yuva420p", "rgba"), ("yuv420p", "x2rgb10le"),
                     ("yuyv422", "yuv420p"), ("gray8", "bgra"), ("yuv420p", "gray8"), ("yuv422p", "uyvy422"), ("yuv420p", "yuva444p"), ("yuv420p", "rgb48le"), ("nv12", "gbrpf32le")):
        for (sw, sh, dw, dh) in ((644, 70, 324, 35), (400, 66, 332, 54), (320, 40, 642, 80), (640, 48, 640, 48), (321, 33, 643, 17)):
            run_case(sw, sh, src, dw, dh, dst, flags, seed=dw, tune=T0)


def test_options_and_slices():
    """chroma positions, range conversion, dither modes and alpha blending next to the flag (the fast functions ignore the horizontal positions: the banks are made
    from xInc alone, hscale_fast_bilinear.c:28-36); source rows handed over in slices"""
    for opts in (dict(dither=1, src_range=1, dst_range=0, src_h_chr_pos=0, src_v_chr_pos=128, dst_h_chr_pos=128, dst_v_chr_pos=0, threads=1),
                 dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1),
                 dict(dither=2, src_range=0, dst_range=0, src_h_chr_pos=256, src_v_chr_pos=0, dst_h_chr_pos=0, dst_v_chr_pos=256, threads=1)):
        for src, dst in (("yuv420p", "yuv420p"), ("yuv420p", "bgra"), ("nv12", "yuv422p"), ("yuvj420p", "yuv420p"), ("yuv420p", "yuvj444p"), ("yuv422p", "rgb565le")):
            for (sw, sh, dw, dh) in ((644, 70, 324, 35), (400, 66, 332, 54), (320, 40, 640, 80), (640, 48, 640, 48)):
                run_case(sw, sh, src, dw, dh, dst, FB | BX, seed=sh, opts=opts, tune=T0)


def test_full_size_frames():
    """the shapes of a player and of a transcoder: 4K -> 1080p, 720p -> 1080p, 1080p same size with a chroma step; HBM and host frames"""
    assert run_case(3840, 2160, "yuv420p", 1920, 1080, "yuv420p", FB, seed=2)[0] == "main:strip_march"
    assert "strip" in run_case(3840, 2160, "yuv420p", 1920, 1080, "bgra", FB, seed=3)[0]
    assert run_case(1280, 720, "yuv420p", 1920, 1080, "yuv420p", FB, seed=4)[0] == "main:strip_march"
    run_case(1280, 720, "nv12", 1920, 1080, "bgra", FB, seed=5)
    run_case(1920, 1080, "yuv422p", 1920, 1080, "yuv420p", FB, seed=6)
    run_case(1920, 1080, "yuv420p", 1920, 1080, "yuv444p", FB | BX, seed=7, device_frames=False)
    run_case(1920, 1080, "yuv420p", 1280, 720, "nv12", FB | BX, seed=8, device_frames=False)
    run_case(1920, 1080, "yuyv422", 1280, 720, "yuv420p", FB, seed=9)
    run_case(3840, 2160, "nv12", 1280, 720, "rgb24", FB, seed=10)
