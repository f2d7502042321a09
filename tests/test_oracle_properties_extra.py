"""CPU properties of the oracle that stand in for reference behaviour no golden vector covers."""
import numpy as np
import pytest

import oracle_lib as OL


@pytest.mark.parametrize("dfmt", ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "gbrp", "gbrap"])
@pytest.mark.parametrize("flags", [4, 4 | 0x80000, 4 | 0x40000, 4 | 0x2000, 2 | 0x2000 | 0x40000, 0x10, 1])
@pytest.mark.parametrize("details", [None, (5, 0, 5, 0, 4096, 1 << 16, 1 << 16), (1, 1, 9, 0, -3000, 70000, 90000)])
def test_gray8_to_byte_rgb_is_the_palette_wrapper(dfmt, flags, details):
    """The reference sends unscaled gray8 -> byte RGB / gbrp / gbrap through palToRgbWrapper / palToGbrpWrapper with the grey palette of
    ff_update_palette (swscale_unscaled.c:600-699, :2619-2630; swscale.c:901-902, usePal swscale_internal.h:937-950): every channel = the
    grey value, A = 255, whatever range / brightness / contrast sws_setColorspaceDetails() was given (round 2 took the scaler chain here,
    which agrees only for the default details)."""
    for (w, h) in ((64, 32), (33, 17)):
        o = OL.Oracle(w, h, "gray8", w, h, dfmt, flags)
        assert o.path() == "palToRgb"
        if details:
            o.set_colorspace(details[0], details[1], details[2], details[3], details[4], details[5], details[6])
        s = OL.Frame("gray8", w, h)
        s.planes[0][:, :w] = (np.arange(w * h) % 256).reshape(h, w).astype(np.uint8)
        d = OL.Frame(dfmt, w, h, fill=0xA5)
        assert o.scale(s, d) >= 0
        g = s.planes[0][:, :w]
        if dfmt in ("gbrp", "gbrap"):
            assert all(np.array_equal(p[:, :w], g) for p in d.planes[:3])
            if dfmt == "gbrap":
                assert (d.planes[3][:, :w] == 255).all()
            continue
        step = 3 if dfmt in ("rgb24", "bgr24") else 4
        v = d.planes[0][:, :w * step].reshape(h, w, step)
        ai = None if step == 3 else (0 if dfmt in ("argb", "abgr") else 3)
        for k in range(step):
            if k == ai:
                assert (v[:, :, k] == 255).all()
            else:
                assert np.array_equal(v[:, :, k], g)
