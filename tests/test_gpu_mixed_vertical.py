"""-m gpu parity for the mixed plan with a vertical chroma step only (identity luma filters, identity horizontal chroma filter): yuv422p / yuvj422p -> yuv420p / nv12
(an MJPEG camera or a 4:2:2 mezzanine into an encoder), yuv420p -> yuv422p, yuv440p <-> yuv444p ..., nv12 / nv16 sources, with and without MPEG <-> JPEG range
conversion, every scaler's vertical bank.  (Round 5 tried a dedicated vertical-filter kernel for the chroma launch -- 16 columns per thread, the tap rows re-read from L2:
parity-green, but no faster than the chroma strip launch, which reads every source row once: 114 vs 120 us on R1 -- and kept the strip launch; the cases stay.)"""
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_AREA, SWS_GAUSS, SWS_SINC, SWS_SPLINE, SWS_BITEXACT, SWS_ACCURATE_RND)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
PATH = "main:plane1+strip_chroma"

PAIRS = [("yuv422p", "yuv420p"), ("yuvj422p", "yuv420p"), ("yuv422p", "yuvj420p"), ("yuv422p", "nv12"), ("yuvj422p", "nv21"), ("yuv420p", "yuv422p"), ("yuvj420p", "yuv422p"), ("yuv420p", "nv16"),
         ("nv16", "yuv420p"), ("nv16", "nv12"), ("nv12", "yuv422p"), ("nv21", "nv16"), ("yuv440p", "yuv444p"), ("yuv444p", "yuv440p"), ("yuvj440p", "yuv444p"), ("yuv422p", "yuv410p"),
         ("yuv420p", "yuvj422p"), ("nv12", "yuvj422p"), ("yuv411p", "yuv410p")]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda c: f"{c[0]}-{c[1]}")
def test_formats_and_scalers(pair):
    src, dst = pair
    for (w, h) in ((640, 48), (1920, 1080), (352, 37), (672, 2), (640, 1), (1376, 50), (3840, 8)):
        for fl in (SWS_BICUBIC, SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_AREA, SWS_GAUSS, SWS_SINC, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND):
            if (w, h) == (1920, 1080) and fl not in (SWS_BICUBIC, SWS_LANCZOS):
                continue
            r = run_case(w, h, src, w, h, dst, fl | BX, seed=w + h)
            if fl == SWS_BICUBIC and w >= 640 and h > 2 and src not in ("yuv411p",):
                assert r[0] == PATH, (r[0], src, dst, w, h)
    run_case(1280, 720, src, 1280, 720, dst, SWS_BICUBIC | BX, seed=5, device_frames=False)


def test_options():
    for opts in (dict(dither=1, src_range=1, dst_range=0, src_h_chr_pos=-513, src_v_chr_pos=128, dst_h_chr_pos=-513, dst_v_chr_pos=0, threads=1),
                 dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1),
                 dict(dither=2, src_range=1, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=0, dst_h_chr_pos=-513, dst_v_chr_pos=256, threads=1)):
        for src, dst in (("yuv422p", "yuv420p"), ("yuv420p", "yuv422p"), ("nv16", "nv12"), ("yuv422p", "nv12")):
            for (w, h) in ((640, 48), (1280, 90), (656, 33)):
                run_case(w, h, src, w, h, dst, SWS_BICUBIC | BX, seed=h, opts=opts)
                run_case(w, h, src, w, h, dst, SWS_LANCZOS | BX, seed=h + 1, opts=opts)
