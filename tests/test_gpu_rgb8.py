"""The 8 / 4 bits-per-pixel RGB destinations: rgb8, bgr8, rgb4_byte, bgr4_byte (one byte per pixel) and rgb4, bgr4 (two pixels per
byte).  LUT writers with the 8x8 ordered-dither tables (yuv2rgb_write "8/4 bits", output.c:1755-1784; tables yuv2rgb.c:817-856),
the unscaled yuv2rgb_c_8 / 4 / 4b_ordered_dither converters (yuv2rgb.c:413-455, :536-559), the full-chroma writers with dither
none / a_dither / x_dither (yuv2rgb_write_full, output.c:2064-2158) and error diffusion, whose error line outlives the frame.
The dither selection rules of utils.c:1288-1316 decide which of them a context gets."""
import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT, SwsContext,
                           HostFrame, DeviceFrame)
from test_gpu_parity import run_case

BX = SWS_BITEXACT
BYTE = ["rgb8", "bgr8", "rgb4_byte", "bgr4_byte"]
NIB = ["rgb4", "bgr4"]
NONE, AUTO, BAYER, ED, A_DITHER, X_DITHER = 0, 1, 2, 3, 4, 5


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", BYTE + NIB)
def test_ordered_dither_lut_writers(fmt):
    # chroma pairs + AUTO -> BAYER: the X, packed2 and packed1 forms of yuv2rgb_{X,2,1}_c_template
    assert run_case(128, 96, "yuv420p", 80, 60, fmt, SWS_BICUBIC | BX, seed=1)[0].startswith("main")
    run_case(64, 48, "yuv420p", 64, 96, fmt, SWS_BILINEAR | BX, seed=2)      # vertical 2-tap up-scale: packed2
    run_case(64, 48, "yuv420p", 40, 48, fmt, SWS_BICUBIC | BX, seed=3)       # vertical identity, 2-tap chroma: packed1 with uvalpha
    run_case(64, 48, "yuv422p", 40, 48, fmt, SWS_POINT | BX, seed=4)         # packed1, uvalpha == 0
    run_case(96, 64, "yuva420p", 50, 38, fmt, SWS_LANCZOS | BX | SWS_ACCURATE_RND, seed=5, device_frames=False)
    run_case(96, 64, "nv12", 52, 38, fmt, SWS_BICUBIC | BX, seed=6, opts=dict(dither=BAYER))
    run_case(96, 64, "p010le", 52, 38, fmt, SWS_BICUBIC | BX, seed=7)
    run_case(96, 64, "yuv444p", 52, 38, fmt, SWS_BICUBIC | BX, seed=8, opts=dict(dither=BAYER))   # BAYER keeps 4:4:4 sources on chroma pairs


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", NIB)
def test_nibble_formats_never_take_full_chroma(fmt):
    # "full chroma interpolation ... not yet implemented" (utils.c:1325-1358): odd widths, 4:4:4 sources and the flag itself all end on
    # the pair writer; the byte of the last pair of an odd width is stored whole
    for w in (61, 63, 1, 2, 3):
        run_case(96, 64, "yuv420p", w, 40, fmt, SWS_BICUBIC | BX, seed=w)
        run_case(96, 64, "yuv420p", w, 40, fmt, SWS_BICUBIC | BX, seed=w, device_frames=False)
    run_case(96, 64, "yuv444p", 60, 40, fmt, SWS_BICUBIC | BX | SWS_FULL_CHR_H_INT, seed=9)
    run_case(96, 64, "rgb24", 60, 40, fmt, SWS_BICUBIC | BX, seed=10, opts=dict(dither=ED))
    run_case(96, 64, "bgra", 60, 40, fmt, SWS_BICUBIC | BX, seed=11, opts=dict(dither=NONE))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", BYTE + NIB)
@pytest.mark.parametrize("src", ["yuv420p", "yuv422p", "yuva420p"])
def test_unscaled_ordered_dither_converters(fmt, src):
    for w, h in ((64, 48), (66, 50), (68, 2), (70, 6), (61, 10), (8, 8), (2, 2), (4, 4), (200, 120)):
        path, opath = run_case(w, h, src, w, h, fmt, SWS_BICUBIC | BX, seed=w + h)
        assert path == "unscaled:yuv2rgb" or (fmt in BYTE and (w & 1)), (path, w)
    run_case(64, 48, src, 64, 48, fmt, SWS_BICUBIC | BX, seed=3, device_frames=False)
    # odd height, accurate_rnd or a dither other than bayer / auto: the scaler chain
    assert run_case(64, 47, src, 64, 47, fmt, SWS_BICUBIC | BX, seed=4)[0] != "unscaled:yuv2rgb"
    assert run_case(64, 48, src, 64, 48, fmt, SWS_BICUBIC | BX | SWS_ACCURATE_RND, seed=5)[0] != "unscaled:yuv2rgb"


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", BYTE)
@pytest.mark.parametrize("dither", [NONE, A_DITHER, X_DITHER])
def test_full_chroma_position_dithers(fmt, dither):
    # these dithers force SWS_FULL_CHR_H_INT (utils.c:1299-1306)
    run_case(128, 96, "yuv420p", 80, 60, fmt, SWS_BICUBIC | BX, seed=1, opts=dict(dither=dither))
    run_case(64, 48, "yuv420p", 64, 96, fmt, SWS_BILINEAR | BX, seed=2, opts=dict(dither=dither))
    run_case(64, 48, "yuv444p", 41, 48, fmt, SWS_BICUBIC | BX, seed=3, opts=dict(dither=dither))
    run_case(64, 48, "yuv420p", 64, 48, fmt, SWS_BICUBIC | BX, seed=4, opts=dict(dither=dither))
    run_case(64, 48, "rgb24", 64, 48, fmt, SWS_BICUBIC | BX, seed=5, opts=dict(dither=dither), device_frames=False)
    run_case(64, 48, "gbrp10le", 50, 40, fmt, SWS_LANCZOS | BX, seed=6, opts=dict(dither=dither))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", BYTE)
def test_error_diffusion(fmt):
    # AUTO with full chroma (odd width, 4:4:4 or RGB source, the flag), BAYER with full chroma, and ED itself all mean error diffusion
    assert run_case(128, 96, "yuv420p", 81, 60, fmt, SWS_BICUBIC | BX, seed=1)[0] == "cascade"
    run_case(128, 96, "yuv444p", 80, 60, fmt, SWS_BICUBIC | BX, seed=2)
    run_case(128, 96, "yuv420p", 80, 60, fmt, SWS_BICUBIC | BX | SWS_FULL_CHR_H_INT, seed=3, opts=dict(dither=BAYER))
    run_case(128, 96, "yuv420p", 80, 60, fmt, SWS_BICUBIC | BX, seed=4, opts=dict(dither=ED))
    run_case(128, 96, "yuv420p", 80, 60, fmt, SWS_BICUBIC | BX, seed=4, opts=dict(dither=ED), device_frames=False)
    run_case(64, 48, "yuv420p", 64, 48, fmt, SWS_BICUBIC | BX, seed=5, opts=dict(dither=ED))          # same size: still the scaler chain
    run_case(64, 48, "rgb24", 64, 48, fmt, SWS_BICUBIC | BX, seed=6)                                   # rgb24 -> rgb8: no shortcut through a copy
    run_case(64, 48, "bgr24", 64, 48, fmt, SWS_BICUBIC | BX, seed=7)
    run_case(64, 48, "rgba", 33, 21, fmt, SWS_BICUBIC | BX, seed=8)
    run_case(33, 1500, "yuv420p", 20, 1100, fmt, SWS_BILINEAR | BX, seed=9)                             # more rows than one wavefront group holds
    run_case(33, 40, "yuv420p", 1, 30, fmt, SWS_BILINEAR | BX, seed=10)
    run_case(33, 40, "yuv420p", 30, 1, fmt, SWS_BILINEAR | BX, seed=11)
    run_case(96, 64, "yuv420p10be", 51, 37, fmt, SWS_BICUBIC | BX, seed=12)
    run_case(96, 64, "yuv420p", 51, 37, fmt, SWS_BICUBIC | BX, seed=13, colorspace=(OL.SWS_CS_ITU709, 1, OL.SWS_CS_ITU601, 0, 3000, 70000, 60000))


@pytest.mark.gpu
@pytest.mark.parametrize("device_frames", [True, False], ids=["hbm", "host"])
@pytest.mark.parametrize("bitexact", [False, True], ids=["carried", "bitexact"])
def test_error_line_outlives_the_frame(device_frames, bitexact):
    """c->dither_error is zeroed at init (utils.c:1744-1747); after that only a bit-exact context clears it, at the start of every frame
    (swscale.c:1084-1086).  Any other context starts the second frame from the first frame's last row."""
    import torch
    sw, sh, dw, dh = 96, 64, 51, 37
    flags = SWS_BICUBIC | (BX if bitexact else 0)
    o = OL.Oracle(sw, sh, "yuv420p", dw, dh, "rgb8", flags)
    p = SwsContext(sw, sh, "yuv420p", dw, dh, "rgb8", flags)
    fresh = None
    for k in range(3):
        src = OL.fill_random(OL.Frame("yuv420p", sw, sh), 5)      # the SAME picture three times
        ref = OL.Frame("rgb8", dw, dh)
        assert o.scale(src, ref) == dh
        hs = HostFrame("yuv420p", sw, sh)
        for a, b in zip(hs.planes, src.planes):
            a[:] = b
        hd = HostFrame("rgb8", dw, dh)
        if device_frames:
            ds, dd = DeviceFrame("yuv420p", sw, sh).upload(hs), DeviceFrame("rgb8", dw, dh)
            torch.cuda.synchronize()
            assert p.scale(ds, dd) == dh
            p.sync()
            dd.download(hd)
        else:
            assert p.scale(hs, hd) == dh
        assert np.array_equal(hd.planes[0][:, :dw], ref.planes[0][:, :dw]), k
        if k == 0:
            fresh = ref.planes[0][:, :dw].copy()
        else:   # ... so without SWS_BITEXACT the same picture does not convert to the same bytes
            assert np.array_equal(ref.planes[0][:, :dw], fresh) == bitexact
    p.close()


@pytest.mark.gpu
def test_error_diffusion_through_the_batch_entry():
    """sws_scale_frames() on an error-diffusion context: the frames are converted in order on one GPU, like a loop of sws_scale()."""
    sw, sh, dw, dh, n = 64, 48, 41, 30, 4
    flags = SWS_BICUBIC
    o = OL.Oracle(sw, sh, "yuv444p", dw, dh, "bgr4_byte", flags)
    p = SwsContext(sw, sh, "yuv444p", dw, dh, "bgr4_byte", flags)
    refs, srcs, dsts = [], [], []
    for k in range(n):
        src = OL.fill_random(OL.Frame("yuv444p", sw, sh), 20 + k)
        ref = OL.Frame("bgr4_byte", dw, dh)
        assert o.scale(src, ref) == dh
        refs.append(ref)
        hs = HostFrame("yuv444p", sw, sh)
        for a, b in zip(hs.planes, src.planes):
            a[:] = b
        srcs.append(hs)
        dsts.append(HostFrame("bgr4_byte", dw, dh))
    assert p.scale_frames(srcs, dsts) == n
    for k in range(n):
        assert np.array_equal(dsts[k].planes[0][:, :dw], refs[k].planes[0][:, :dw]), k
    p.close()


def test_dither_rules_and_refusals(hiplib):
    L = hiplib
    for f in BYTE + NIB:
        assert L.sws_isSupportedOutput(LA.PIX_FMT[f]) == 1
    for f in NIB:   # the bit-stream formats are outputs only (format.c legacy_format_entries); the byte formats are read through a palette (test_gpu_pal.py)
        for make in (OL.Oracle, SwsContext):
            with pytest.raises(RuntimeError):
                make(64, 48, f, 64, 48, "yuv420p", SWS_BICUBIC | BX)
    # utils.c:1293-1316 as seen through the context's options after init
    def after(fmt, sfmt, w, flags, dither):
        c = SwsContext(64, 48, sfmt, w, 40, fmt, flags | BX, dither=dither)
        f = c.fields()
        r = (f.dither, bool(f.flags & SWS_FULL_CHR_H_INT))
        c.close()
        return r
    assert after("rgb8", "yuv420p", 40, SWS_BICUBIC, AUTO) == (BAYER, False)
    assert after("rgb8", "yuv420p", 41, SWS_BICUBIC, AUTO) == (ED, True)
    assert after("rgb8", "yuv444p", 40, SWS_BICUBIC, AUTO) == (ED, True)
    assert after("rgb8", "yuv444p", 40, SWS_BICUBIC, BAYER) == (BAYER, False)
    assert after("bgr4_byte", "yuv420p", 40, SWS_BICUBIC, A_DITHER) == (A_DITHER, True)
    assert after("bgr4_byte", "yuv420p", 40, SWS_BICUBIC, NONE) == (NONE, True)
    assert after("bgr4_byte", "yuv420p", 40, SWS_BICUBIC | SWS_FULL_CHR_H_INT, BAYER) == (ED, True)
    assert after("rgb4", "yuv420p", 41, SWS_BICUBIC | SWS_FULL_CHR_H_INT, AUTO) == (AUTO, False)
    assert after("rgb8", "yuv420p", 40, SWS_BICUBIC | (1 << 23), AUTO) == (ED, True)   # SWS_ERROR_DIFFUSION
