"""The input-only formats of the reference's table (format.c legacy_format_entries): packed float RGB (rgbf32, rgbf16, rgbaf16), float
gray (grayf16, yaf32, yaf16), planar half-float RGB (gbrpf16, gbrapf16), LE and BE, and the packed 4:1:1 layout uyyvyy411.  Every
element is converted with lrintf(av_clipf(65535.0f * x, 0, 65535)) (half-floats through the exact half2float widening) and then runs
through the 16-bit RGB arithmetic: input.c:909-925, :1336-1431, :1558-1740.  tests/oracle_lib.fill_random() sprinkles negative values,
values above 1, infinities, NaNs and (for half-floats) arbitrary bit patterns, subnormals included."""
import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_FAST_BILINEAR, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INP,
                           SWS_FULL_CHR_H_INT, SwsContext)
from test_gpu_parity import run_case, FLOAT_IN

BX = SWS_BITEXACT
DSTS = ["yuv420p", "yuv444p16le", "rgb24", "bgra", "rgba64le", "gbrp", "gbrapf32le", "gray8", "gray16be", "grayf32le", "ya8", "ya16le", "yuva420p", "yuva444p10le",
        "nv12", "p010le", "rgb565le", "monob", "rgb8"]


@pytest.mark.gpu
@pytest.mark.parametrize("src", FLOAT_IN)
@pytest.mark.parametrize("dst", DSTS)
def test_float_and_411_sources(src, dst):
    run_case(96, 64, src, 60, 40, dst, SWS_BICUBIC | BX, seed=1)                           # even width: RGB chroma from averaged pixel pairs
    run_case(97, 64, src, 60, 41, dst, SWS_BILINEAR | BX, seed=2)                          # odd width: chroma per pixel
    run_case(96, 64, src, 60, 40, dst, SWS_LANCZOS | BX | SWS_FULL_CHR_H_INP | SWS_ACCURATE_RND, seed=3, device_frames=False)
    run_case(64, 40, src, 64, 40, dst, SWS_BICUBIC | BX, seed=4)                           # same size: no special converter takes these sources
    run_case(64, 40, src, 128, 80, dst, SWS_FAST_BILINEAR | BX, seed=5)
    run_case(64, 40, src, 50, 30, dst, SWS_POINT | BX | (1 << 16), seed=6)                 # SWS_SRC_V_CHR_DROP 1


def test_input_only(hiplib):
    L = hiplib
    for f in FLOAT_IN:
        assert L.sws_isSupportedInput(LA.PIX_FMT[f]) == 1 and L.sws_isSupportedOutput(LA.PIX_FMT[f]) == 0, f
        for make in (OL.Oracle, SwsContext):
            with pytest.raises(RuntimeError):
                make(64, 48, "yuv420p", 64, 48, f, SWS_BICUBIC | BX)


@pytest.mark.gpu
def test_float_gray_keeps_the_range_it_was_given():
    """handle_jpeg (utils.c:773-809) names gray8 .. gray16 and ya8 / ya16, not the float gray formats: a grayf32 / grayf16 / yaf picture is
    limited-range unless the caller says otherwise, so grayf32 -> gray8 stretches 16..235 to 0..255 and grayf32 -> yuv420p does not."""
    W, H = 64, 16
    for fmt, dt in (("grayf32le", np.float32), ("grayf16le", np.float16)):
        src = OL.Frame(fmt, W, H)
        src.planes[0][:, :W * np.dtype(dt).itemsize] = np.full((H, W), 16 / 255.0, dtype=dt).view(np.uint8)
        for dst, want in (("gray8", 0), ("yuv420p", 16)):
            o = OL.Oracle(W, H, fmt, W // 2, H, dst, SWS_BICUBIC | BX)
            out = OL.Frame(dst, W // 2, H)
            assert o.scale(src, out) == H
            assert int(out.planes[0][0, 0]) == want, (fmt, dst)
        run_case(W, H, fmt, W // 2, H, "gray8", SWS_BICUBIC | BX, seed=1)
        run_case(W, H, fmt, W // 2, H, "gray8", SWS_BICUBIC | BX, seed=1, opts=dict(src_range=1))
        run_case(W, H, fmt, W // 2, H, "yuv420p", SWS_BICUBIC | BX, seed=2, opts=dict(src_range=1, dst_range=1))
