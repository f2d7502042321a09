"""The extreme-ratio cascade (libswscale/utils.c:1803-1833): a filter of 256 taps or more makes sws_init_context() build two
contexts around a yuv420p / yuva420p picture of the geometric-mean size (scale_cascaded, swscale.c:992-1018)."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SwsContext
from test_gpu_parity import run_case

BX = SWS_BITEXACT

CASES = [
    (2048, 1200, "yuv420p", 24, 16, "yuv420p", SWS_BICUBIC),          # both directions need the cascade
    (2048, 64, "yuv420p", 24, 48, "rgb24", SWS_BICUBIC),              # only the horizontal filter does
    (64, 2100, "nv12", 48, 30, "bgra", SWS_BICUBIC | SWS_ACCURATE_RND),   # only the vertical one
    (1500, 900, "rgba", 20, 14, "yuva420p", SWS_LANCZOS),             # alpha: yuva420p in between
    (1500, 900, "yuv422p10le", 20, 14, "p010le", SWS_LANCZOS),
    (4100, 40, "gray8", 30, 20, "gray8", SWS_BILINEAR),               # bilinear: 128:1
    (2048, 1200, "yuv420p16be", 24, 16, "yuv420p10be", SWS_BICUBIC),  # the byte order travels to the first / second step
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_host_side_builds_the_same_cascade(hiplib, case):
    """CPU: the product and the oracle agree on when the cascade is needed"""
    sw, sh, sf, dw, dh, df, flags = case
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags | BX)
    p = SwsContext(sw, sh, sf, dw, dh, df, flags | BX)
    assert o.path() == "cascade" and hiplib.sws_hip_path_name(p.c) == b"cascade"
    p.close()


def test_moderate_ratios_do_not_cascade(hiplib):
    o = OL.Oracle(2048, 1200, "yuv420p", 64, 40, "yuv420p", SWS_BICUBIC | BX)
    assert o.path() == "main"
    # a cascade request with too small an area ratio is an error in the reference (utils.c:1811-1812)
    with pytest.raises(RuntimeError):
        OL.Oracle(4100, 16, "gray8", 30, 600, "gray8", SWS_BILINEAR | BX)
    with pytest.raises(RuntimeError):
        SwsContext(4100, 16, "gray8", 30, 600, "gray8", SWS_BILINEAR | BX)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_extreme_ratio_parity(case):
    sw, sh, sf, dw, dh, df, flags = case
    path, opath = run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=5)
    assert (path, opath) == ("cascade", "cascade")
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=6, device_frames=False)
