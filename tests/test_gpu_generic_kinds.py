"""-m gpu: the element-per-thread kernels exist twice -- one instantiation per source kind / destination kind (k_generic_kinds.hip, k_generic_dst.hip:
the default) and the all-kinds forms of k_generic.hip (option no_generic_kinds = 1).  The format matrices of test_gpu_parity.py run the default; this
file runs one source format of every SrcKind against one destination format of every DstKind through BOTH, same size (the single-pass kernels) and
scaled (pass 1 + pass 2, or the fused tile kernel), each against the oracle."""
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BITEXACT, SWS_ACCURATE_RND
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT

SRC = ["yuv420p", "yuv422p10le", "yuv444p16le", "nv12", "p010le", "p016le", "rgb24", "bgra", "gbrp", "gbrap", "gbrpf32le", "rgb48le", "rgba64le", "yuyv422", "gbrp10le", "gbrap16le",
       "y210le", "xv30le", "xv36le", "ayuv64le", "ayuv", "vuya", "vyu444", "ya8", "ya16le", "grayf32le", "monob", "x2rgb10le", "rgbf32le", "rgbaf16le", "gbrpf16le", "grayf16le",
       "pal8", "rgb8", "uyyvyy411", "rgb565le", "bgr444le", "gray10le", "yuva420p"]
DST = ["yuv420p", "yuv422p10le", "yuv444p16le", "nv12", "p010le", "rgb24", "bgra", "gbrp", "gbrp10le", "gbrap16le", "gbrpf32le", "rgb48le", "rgba64le", "uyvy422", "p016le", "y210le",
       "xv30le", "ayuv64le", "ayuv", "ya8", "ya16le", "grayf32le", "monow", "x2rgb10le", "rgb8", "rgb4", "bgr4_byte", "rgb565le", "gray8", "yuva420p"]


def _both(sw, sh, sfmt, dw, dh, dfmt, flags, seed):
    from librempeg_amd import SwsContext
    try:
        OL.Oracle(sw, sh, sfmt, dw, dh, dfmt, flags)
    except RuntimeError:
        pytest.skip("oracle does not restate this converter")
    try:
        SwsContext(sw, sh, sfmt, dw, dh, dfmt, flags).close()
    except RuntimeError:
        pytest.skip("not implemented (sws_getContext -> NULL)")
    path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, flags, seed=seed)
    if path.startswith("main:") and any(k in path for k in ("generic", "two_pass", "fused_tile")):
        path2, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, flags, seed=seed, tune={"no_generic_kinds": 1})
        assert path2 == path


@pytest.mark.parametrize("sfmt", SRC)
@pytest.mark.parametrize("dfmt", DST)
def test_kinds_same_size(sfmt, dfmt):
    _both(70, 38, sfmt, 70, 38, dfmt, SWS_BICUBIC | BX, 11)


@pytest.mark.parametrize("sfmt", SRC)
@pytest.mark.parametrize("dfmt", DST)
def test_kinds_scaled(sfmt, dfmt):
    _both(90, 50, sfmt, 58, 34, dfmt, SWS_BICUBIC | BX | SWS_ACCURATE_RND, 12)


@pytest.mark.parametrize("sfmt,dfmt", [("bgra", "y210le"), ("yuv420p", "rgba64le"), ("gbrpf32le", "bgra"), ("ayuv", "bgra"), ("rgb565le", "yuv420p"), ("xv30le", "nv12"),
                                       ("rgba64le", "yuva420p"), ("ya8", "bgra"), ("yuv420p10le", "x2rgb10le")])
@pytest.mark.parametrize("geo", [(642, 362, 642, 362), (642, 362, 322, 182), (322, 182, 642, 362)])
def test_kinds_larger_pictures(sfmt, dfmt, geo):
    """pictures of several blocks per row: ragged last block, odd widths"""
    sw, sh, dw, dh = geo
    _both(sw, sh, sfmt, dw, dh, dfmt, SWS_BICUBIC | BX, 13)


@pytest.mark.parametrize("geo", [(130, 40, 130, 40), (130, 40, 86, 26)])
@pytest.mark.parametrize("sfmt,dfmt", [("bgra", "ayuv"), ("argb", "vuya"), ("rgba", "uyva")])
@pytest.mark.parametrize("pad,shift", [(1, 1), (3, 3), (2, 2)])
def test_rgb32_reader_on_unaligned_pixels(sfmt, dfmt, geo, pad, shift):
    """the 32 bpp RGB reader loads a pixel as one dword: frames whose pixels are not 4-byte aligned (a view into a byte buffer) must read the same"""
    from test_gpu_unaligned_frames import run_odd
    sw, sh, dw, dh = geo
    run_odd(sw, sh, sfmt, dw, dh, dfmt, SWS_BICUBIC | BX, pad, shift, 0)
