"""Gamma-correct scaling (SwsContext.gamma_flag, libswscale/utils.c:1461-1522, gamma.c, swscale.c:959-990): source -> RGBA64LE, scaled
between a pow(x, 1/2.2) and a pow(x, 2.2) table pass, -> destination.  Oracle and product are two restatements (no reference golden
exercises this option)."""
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SwsContext
from test_gpu_parity import run_case

BX = SWS_BITEXACT
CASES = [
    (96, 64, "rgb24", 64, 40, "rgb24", SWS_BICUBIC), (96, 64, "yuv420p", 128, 80, "yuv420p", SWS_BILINEAR),
    (96, 64, "rgba64le", 64, 40, "rgba64le", SWS_BICUBIC),        # the reference builds no cascade (utils.c:1465) and so applies no gamma at all
    (96, 64, "rgba64le", 64, 40, "bgra", SWS_LANCZOS), (96, 64, "nv12", 64, 40, "rgba64le", SWS_BICUBIC),
    (96, 64, "rgba", 61, 37, "yuva420p", SWS_BICUBIC | SWS_ACCURATE_RND), (96, 64, "rgba64be", 64, 40, "rgba64be", SWS_BICUBIC),
    (96, 64, "yuv420p10le", 64, 40, "gbrp10le", SWS_BICUBIC), (1280, 72, "yuv420p", 640, 36, "rgb24", SWS_BICUBIC),
]


def test_same_size_ignores_the_flag(hiplib):
    o = OL.Oracle(64, 32, "yuv420p", 64, 32, "rgb24", SWS_BICUBIC | BX, gamma_flag=1)
    assert o.path() != "cascade"


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_gamma_cascade_parity(case):
    sw, sh, sf, dw, dh, df, flags = case
    path, opath = run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=3, opts=dict(gamma_flag=1))
    want = ("main:two_pass", "main") if sf == df == "rgba64le" else ("cascade", "cascade")
    assert (path, opath) == want
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=4, opts=dict(gamma_flag=1), device_frames=False)
