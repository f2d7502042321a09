"""SWS_SRC_V_CHR_DROP (swscale.h:453-454, utils.c:1362-1365, swscale.c:333-334): the scaler reads every 2nd / 4th / 8th row of the
source chroma planes.  Oracle and product against each other over planar, semi-planar, packed and RGB sources, every kernel family
of the scaled path (strip, strip + RGB epilogue, tile, two-pass), whole frames and slices."""
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT


def drop(n):
    return n << 16


CASES = [
    (96, 64, "yuv420p", 64, 40, "yuv420p", SWS_BICUBIC), (96, 64, "yuv420p", 96, 64, "yuv420p", SWS_BICUBIC),
    (96, 64, "yuv422p", 128, 80, "nv12", SWS_LANCZOS), (96, 64, "yuv444p", 64, 48, "yuv420p", SWS_BILINEAR),
    (96, 64, "nv12", 64, 40, "rgb24", SWS_BICUBIC), (96, 64, "yuv420p", 64, 40, "bgra", SWS_BICUBIC | SWS_ACCURATE_RND),
    (96, 64, "yuv420p", 96, 64, "rgb24", SWS_BICUBIC | SWS_ACCURATE_RND), (96, 64, "yuyv422", 64, 40, "yuv420p", SWS_BICUBIC),
    (96, 64, "yuv420p10le", 48, 32, "p010le", SWS_LANCZOS), (96, 64, "rgb24", 64, 40, "yuv420p", SWS_BICUBIC),
    (2048, 40, "yuv420p", 1024, 24, "yuv420p", SWS_BICUBIC), (1100, 48, "yuv420p", 550, 24, "rgb24", SWS_BICUBIC),
    (61, 33, "yuv440p", 47, 29, "yuv422p", SWS_BICUBIC), (96, 64, "uyvy422", 96, 64, "nv12", SWS_BICUBIC), (96, 64, "vuya", 64, 40, "yuv420p", SWS_BICUBIC),
    (96, 64, "y210le", 64, 40, "yuv422p10le", SWS_BICUBIC), (96, 64, "gbrp", 64, 40, "yuv420p", SWS_BICUBIC), (96, 64, "bgra", 64, 40, "yuv420p", SWS_BICUBIC),
    (96, 64, "rgb565le", 64, 40, "yuv420p", SWS_BICUBIC), (96, 64, "rgb48le", 64, 40, "yuv420p", SWS_BICUBIC), (96, 64, "gbrp10le", 64, 40, "yuv420p10le", SWS_BICUBIC),
]


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_vchrdrop(case, n):
    sw, sh, sf, dw, dh, df, flags = case
    run_case(sw, sh, sf, dw, dh, df, flags | BX | drop(n), seed=7 + n)
    run_case(sw, sh, sf, dw, dh, df, flags | BX | drop(n), seed=9 + n, device_frames=False)
