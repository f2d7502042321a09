"""Error diffusion into the 1 bpp formats: yuv2mono_{X,2,1}_c_template with SWS_DITHER_ED (output.c:690-700, :734-753, :792-811).
Floyd-Steinberg over the luma with threshold 128 and step 220; the pair loop runs a phantom pixel through the recurrence for an odd
width; the X form stores a trailing partial byte, the 2 and 1 forms leave it alone; the error line lives as long as the context
unless SWS_BITEXACT clears it per frame (swscale.c:1084-1086)."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_BITEXACT, SWS_ACCURATE_RND, SwsContext, HostFrame, DeviceFrame
from test_gpu_parity import run_case

BX = SWS_BITEXACT
ED = dict(dither=3)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["monob", "monow"])
@pytest.mark.parametrize("src", ["yuv420p", "gray8", "rgb24", "yuv444p10le", "monob"])
def test_mono_error_diffusion(fmt, src):
    for dw in (64, 61, 45, 8, 7, 2, 1):
        assert run_case(96, 64, src, dw, 40, fmt, SWS_BICUBIC | BX, seed=dw, opts=ED)[0] == "cascade"     # X form
        run_case(96, 32, src, dw, 64, fmt, SWS_BILINEAR | BX, seed=dw + 1, opts=ED)                        # vertical 2-tap: the 2 form on most rows
        run_case(96, 40, src, dw, 40, fmt, SWS_POINT | BX, seed=dw + 2, opts=ED, device_frames=False)      # the 1 form
    run_case(64, 48, src, 64, 48, fmt, SWS_BICUBIC | BX, seed=3, opts=ED)                                  # same size: the scaler chain (the unscaled 1 bpp converter is ordered dither only)
    run_case(64, 48, src, 64, 48, fmt, SWS_BICUBIC | BX | (1 << 23), seed=4)                               # SWS_ERROR_DIFFUSION turns AUTO into ED
    run_case(40, 1500, src, 21, 1100, fmt, SWS_BILINEAR | BX, seed=5, opts=ED)                             # more rows than one wavefront group


@pytest.mark.gpu
@pytest.mark.parametrize("bitexact", [False, True], ids=["carried", "bitexact"])
def test_mono_error_line_between_frames(bitexact):
    sw, sh, dw, dh = 96, 64, 53, 37
    flags = SWS_LANCZOS | (BX if bitexact else 0)
    o = OL.Oracle(sw, sh, "yuv420p", dw, dh, "monow", flags, **ED)
    p = SwsContext(sw, sh, "yuv420p", dw, dh, "monow", flags, **ED)
    first = None
    nb = (dw + 7) // 8
    for k in range(3):
        src = OL.fill_random(OL.Frame("yuv420p", sw, sh), 5)
        ref = OL.Frame("monow", dw, dh)
        assert o.scale(src, ref) == dh
        hs = HostFrame("yuv420p", sw, sh)
        for a, b in zip(hs.planes, src.planes):
            a[:] = b
        hd = HostFrame("monow", dw, dh)
        assert p.scale(hs, hd) == dh
        m = (0xFF00 >> (dw & 7)) & 0xFF if dw & 7 else 0xFF
        a, b = hd.planes[0][:, :nb].copy(), ref.planes[0][:, :nb].copy()
        a[:, nb - 1] &= m; b[:, nb - 1] &= m
        assert np.array_equal(a, b), k
        if k == 0:
            first = b
        else:
            assert np.array_equal(b, first) == bitexact
    p.close()
