"""alpha_blend (SwsContext.alpha_blend; libswscale/utils.c:1565-1616, alphablend.c): a source with an alpha channel converted to a
destination without one is first blended over a uniform (black) or checkerboard background -- directly (ff_sws_alphablendaway as the
special converter) when the destination is the alpha-less twin of the source format at the same size, through a two-step cascade
otherwise."""
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SwsContext
from test_gpu_parity import run_case

BX = SWS_BITEXACT
DIRECT = [("rgba", "rgb24"), ("argb", "rgb24"), ("bgra", "bgr24"), ("abgr", "bgr24"), ("ya8", "gray8"), ("yuva420p", "yuv420p"),
          ("yuva422p", "yuv422p"), ("yuva444p", "yuv444p"), ("rgba64le", "rgb48le"), ("bgra64le", "bgr48le"), ("ya16le", "gray16le"),
          ("yuva420p10le", "yuv420p10le"), ("yuva422p9le", "yuv422p9le"), ("yuva444p16le", "yuv444p16le"), ("rgba64be", "rgb48le"),
          ("yuva420p16be", "yuv420p16le")]
CASCADE = [(97, 65, "rgba", 64, 40, "yuv420p", SWS_BICUBIC), (97, 65, "yuva420p", 97, 65, "rgb24", SWS_BICUBIC),
           (96, 64, "yuva444p10le", 128, 80, "nv12", SWS_LANCZOS), (96, 64, "ya8", 96, 64, "rgb24", SWS_BICUBIC),
           (96, 64, "rgba64be", 96, 64, "rgb48be", SWS_BICUBIC), (96, 64, "bgra", 48, 32, "gray8", SWS_BILINEAR | SWS_ACCURATE_RND)]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2], ids=["uniform", "checkerboard"])
@pytest.mark.parametrize("pair", DIRECT, ids=lambda p: f"{p[0]}-{p[1]}")
def test_blendaway_special_converter(pair, mode):
    sf, df = pair
    for (w, h) in ((97, 67), (64, 34)):
        path, opath = run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=w, opts=dict(alpha_blend=mode))
        assert (path, opath) == ("unscaled:alphablendaway", "alphablendaway")
        run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=h, opts=dict(alpha_blend=mode), device_frames=False)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2], ids=["uniform", "checkerboard"])
@pytest.mark.parametrize("case", CASCADE, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_blend_cascade(case, mode):
    sw, sh, sf, dw, dh, df, flags = case
    path, opath = run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=11, opts=dict(alpha_blend=mode))
    assert (path, opath) == ("cascade", "cascade")


def test_without_the_option_alpha_is_dropped(hiplib):
    o = OL.Oracle(64, 32, "rgba", 64, 32, "rgb24", SWS_BICUBIC | BX)
    assert o.path() == "rgbToRgb"
    # formats without an alpha-less twin in the reference's table are not blended (utils.c:1113-1117)
    o = OL.Oracle(64, 32, "yuva444p12le", 64, 32, "yuv444p12le", SWS_BICUBIC | BX, alpha_blend=1)
    assert o.path() != "alphablendaway" and o.path() != "cascade"
