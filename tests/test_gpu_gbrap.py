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


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("gbrap", "rgba"), ("gbrap", "bgra"), ("gbrap", "argb"), ("gbrap", "abgr"), ("gbrap", "rgb24"), ("gbrap", "bgr24"), ("gbrap", "bgr0"),
                                  ("rgba", "gbrap"), ("bgra", "gbrap"), ("argb", "gbrap"), ("abgr", "gbrap"), ("rgb24", "gbrap"), ("bgr24", "gbrap"), ("rgb0", "gbrap"), ("0bgr", "gbrap"),
                                  ("rgba64le", "gbrap10le"), ("bgra64le", "gbrap16le"), ("rgb48le", "gbrap12le"), ("bgr48be", "gbrap14be"), ("rgba64be", "gbrap16le"), ("rgba64le", "gbrp12le"),
                                  ("gbrap16le", "rgb48le"), ("gbrap10le", "rgba64le"), ("gbrap12be", "bgra64le"), ("gbrap14le", "bgra64be"), ("gbrp10le", "rgba64le"), ("gbrap16le", "bgr48be"),
                                  ("x2rgb10le", "gbrap12le"), ("x2bgr10le", "gbrap10le"), ("x2rgb10le", "gbrap16be"), ("gbrap10le", "x2rgb10le"), ("gbrap16le", "x2bgr10le")],
                         ids=lambda p: f"{p[0]}-{p[1]}")
def test_alpha_rows_of_the_packed_planar_wrappers(pair):
    """planarRgbaToRgbWrapper / gbraptopacked32 (swscale_unscaled.c:1235-1320), rgbToPlanarRgbaWrapper / packed24togbrap / packed32togbrap
    (:1480-1590), packed16togbra16 and gbr16ptopacked16 with an alpha plane on either side (:685-817, :964-1081), packed30togbra10's
    all-ones alpha (:819-889)."""
    for w, h in ((70, 38), (1, 1), (129, 5)):
        path, opath = run_case(w, h, pair[0], w, h, pair[1], SWS_BICUBIC | BX, seed=w)
        assert path.startswith("unscaled:") and opath != "main", (path, opath)
    run_case(70, 38, pair[0], 70, 38, pair[1], SWS_BICUBIC | BX, seed=9, device_frames=False)


@pytest.mark.gpu
def test_msb_planar_rgb_is_not_in_the_16_bit_wrapper_rules():
    # Rgb16ToPlanarRgb16Wrapper / planarRgb16ToRgb16Wrapper name gbrp9..16 and gbrap10..16 (:2495-2533): gbrp10msb / gbrp12msb go through the scaler
    for s, d in (("rgb48le", "gbrp10msble"), ("gbrp12msble", "rgba64le"), ("bgra64le", "gbrp12msbbe")):
        assert run_case(70, 38, s, 70, 38, d, SWS_BICUBIC | BX, seed=3)[1] == "main"
    # ... the 30 bpp rules take every planar RGB format of 10 bits and more (:2509-2512, :2535-2538)
    assert OL.Oracle(70, 38, "x2rgb10le", 70, 38, "gbrp10msble", SWS_BICUBIC | BX).path() != "main"
    run_case(70, 38, "x2rgb10le", 70, 38, "gbrp10msble", SWS_BICUBIC | BX, seed=4)
    run_case(70, 38, "gbrp12msble", 70, 38, "x2bgr10le", SWS_BICUBIC | BX, seed=5)
