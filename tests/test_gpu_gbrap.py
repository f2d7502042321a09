"""The planar RGB + alpha family (gbrap, gbrap10/12/14/16 LE and BE, gbrapf32 LE and BE): planar_rgb*_to_a readers (input.c:1188-1298),
the hasAlpha arms of yuv2gbrp_full_X_c / yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c (output.c:2343-2605), the alpha fill of
swscale.c:536-552 incl. fillPlane32, planarCopyWrapper over four planes and planarRgbToplanarRgbWrapper (gbrp <-> gbrap).  The four
fate-sws-floatimg-cmp lines through these formats are reproduced by tests/test_oracle_floatimg.py and tests/test_gpu_floatimg.py."""
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SwsContext
from test_gpu_parity import run_case

BX = SWS_BITEXACT
GBRAP = ["gbrap", "gbrap10le", "gbrap12le", "gbrap14le", "gbrap16le", "gbrapf32le", "gbrap10be", "gbrap16be", "gbrapf32be"]
OTHERS = ["rgba", "yuva420p", "yuv420p", "rgb24", "bgra64le", "gbrp", "gbrp12le", "gbrpf32le", "yuva444p16le", "ya8", "gray8", "nv12", "p010le"]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", GBRAP)
@pytest.mark.parametrize("other", OTHERS)
def test_scaled_to_and_from(fmt, other):
    run_case(96, 64, other, 64, 40, fmt, SWS_BICUBIC | BX, seed=1)
    run_case(96, 64, fmt, 61, 37, other, SWS_BILINEAR | BX, seed=2)
    run_case(64, 40, other, 64, 40, fmt, SWS_BICUBIC | BX | SWS_ACCURATE_RND, seed=3, device_frames=False) if _built(other, fmt) else None
    run_case(64, 40, fmt, 64, 40, other, SWS_BICUBIC | BX, seed=4) if _built(fmt, other) else None


def _built(s, d):
    try:
        OL.Oracle(64, 40, s, 64, 40, d, SWS_BICUBIC | BX)
        return True
    except RuntimeError:
        return False


@pytest.mark.gpu
@pytest.mark.parametrize("a", GBRAP)
@pytest.mark.parametrize("b", GBRAP)
def test_within_the_family(a, b):
    run_case(70, 38, a, 70, 38, b, SWS_BICUBIC | BX, seed=5)
    run_case(70, 38, a, 50, 30, b, SWS_LANCZOS | BX, seed=6)


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("gbrp", "gbrap"), ("gbrap", "gbrp"), ("gbrp10le", "gbrap10le"), ("gbrap12le", "gbrp12le"), ("gbrp14le", "gbrap14le"),
                                  ("gbrap16le", "gbrp16le")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_planar_rgb_to_planar_rgb_wrapper(pair):
    path, opath = run_case(70, 38, pair[0], 70, 38, pair[1], SWS_BICUBIC | BX, seed=7)
    assert (path, opath) == ("unscaled:planarRgbToplanarRgb", "planarRgbToplanarRgb")
    run_case(70, 38, pair[0], 70, 38, pair[1], SWS_BICUBIC | BX, seed=8, device_frames=False)


def test_wrappers_that_are_not_built_are_refused_by_both(hiplib):
    for s, d in (("gbrap", "rgba"), ("rgba", "gbrap"), ("rgba64le", "gbrap10le"), ("gbrap16le", "rgb48le"), ("x2rgb10le", "gbrap12le")):
        with pytest.raises(RuntimeError):
            OL.Oracle(64, 32, s, 64, 32, d, SWS_BICUBIC | BX)
        with pytest.raises(RuntimeError):
            SwsContext(64, 32, s, 64, 32, d, SWS_BICUBIC | BX)
        # ... scaled, the same pair goes through the scaler
        OL.Oracle(64, 32, s, 48, 24, d, SWS_BICUBIC | BX)
        SwsContext(64, 32, s, 48, 24, d, SWS_BICUBIC | BX).close()
