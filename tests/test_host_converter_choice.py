"""CPU box: which of the reference's special converters a conversion at equal sizes takes (swscale_unscaled.c ff_get_unscaled_swscale :2389-2713, the alpha-blend and
cascade rules of utils.c) -- the PRODUCT's planner (dry_plan, no GPU) against the ORACLE's, for every ordered pair of the 234 formats under eight configurations (flag sets, chroma drop, range change, alpha blending, gamma, three scaled sizes).  The oracle's
choice is held to the real reference by outputs (tests/test_oracle_reference_answers_r06.py, tools/ref/ref_crosscheck.py); the GPU suite compares pixels on a sample of
pairs; this test is the full matrix of the rule table itself, the place where round 6 found three rules both sides had wrong in the same way and one the product still
declines by design (bswap_16bpc with SWS_SRC_V_CHR_DROP: refused, never a wrong picture)."""
import ctypes as C

import pytest

import oracle_lib as OL
from librempeg_amd import swscale as S
from librempeg_amd import (SWS_ACCURATE_RND, SWS_BICUBIC, SWS_BILINEAR, SWS_BITEXACT, SWS_FAST_BILINEAR, SWS_FULL_CHR_H_INT, SWS_LANCZOS, SWS_POINT)

FORMATS = sorted(S._FORMATS)
# (flags, options, destination size of a 64x36 source)
CONFIGS = [(SWS_BICUBIC | SWS_BITEXACT, {}, (64, 36)), (SWS_FAST_BILINEAR | (1 << 16), {}, (64, 36)),          # (1 << 16: SWS_SRC_V_CHR_DROP of one)
           (SWS_POINT | SWS_ACCURATE_RND | SWS_FULL_CHR_H_INT, {}, (64, 36)), (SWS_BILINEAR | SWS_BITEXACT, dict(alpha_blend=1), (64, 36)),
           (SWS_BICUBIC, dict(src_range=1, dst_range=0), (64, 36)), (SWS_BICUBIC | SWS_BITEXACT, dict(alpha_blend=2), (96, 20)),
           (SWS_LANCZOS | SWS_FULL_CHR_H_INT | SWS_BITEXACT, {}, (640, 4)), (SWS_BILINEAR, dict(gamma_flag=1), (32, 72))]
SAME = {"yuv2rgb_c": "yuv2rgb"}


def _product(L, sf, df, flags, opts, dw, dh):
    try:
        p = S.SwsContext(64, 36, sf, dw, dh, df, flags, **opts)
    except Exception:
        return None
    try:
        p.set_option("dry_plan", 1)
        r = L.sws_hip_plan(p.c, (C.c_uint64 * 3)())
        return p.path() if r >= 0 else f"plan error {r}"
    finally:
        p.close()


@pytest.mark.parametrize("sf", FORMATS)
def test_converter_choice(sf):
    L = S.load_library()
    L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    bad = []
    for flags, opts, (dw, dh) in CONFIGS:
        for df in FORMATS:
            try:
                want = OL.Oracle(64, 36, sf, dw, dh, df, flags, **opts).path()
            except Exception:
                want = None
            got = _product(L, sf, df, flags, opts, dw, dh)
            if want is None or got is None:
                ok = want is None and got is None
            elif want == "bswap_16bpc":
                ok = False                                                  # (the product refuses these: got is None, handled above)
            elif want in ("main", "cascade"):
                ok = got.startswith("main:") or got.startswith("cascade")   # (the product splits some scaler conversions into steps of its own)
                ok = ok and (want != "cascade" or got.startswith("cascade"))
            else:
                ok = got.startswith("unscaled:") and got.split(":")[1].split("+")[0] == SAME.get(want, want)
            if want == "bswap_16bpc" and got is None:
                ok = True
            if not ok:
                bad.append((sf, df, hex(flags), opts, (dw, dh), want, got))
    assert not bad, bad[:20]
