"""-m gpu parity at widths of 4 k + 2 on either side (1366 x 768 screens, 854 x 480, 1282-wide windows): round 5 took the multiple-of-4 rules off the reader pre-pass of RGB
sources and off the packed destinations behind the sum planes (rgb565 family, x2rgb10, packed 4:4:4 / 10-bit packed YUV, rgb48 / rgba64, planar RGB of 16 bits / float32)."""
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_AREA, SWS_BITEXACT
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
SRC = ["yuv420p", "nv12", "yuv420p10le", "bgra", "rgb24", "yuv444p", "yuv422p", "p010le", "gray8"]
DST = ["rgb565le", "bgr555le", "x2rgb10le", "ayuv", "vuya", "vyu444", "xv30le", "y210le", "xv36le", "rgb48le", "rgba64le", "bgr48be", "gbrp16le", "gbrpf32le", "gbrp10le", "bgra", "yuyv422", "p016le",
       "yuv420p16le", "rgb444le"]
GEOM = [(854, 48, 1282, 72), (1366, 40, 1282, 38), (642, 30, 322, 15), (1280, 36, 1366, 38), (646, 26, 646, 26), (1918, 22, 1278, 14), (322, 60, 1286, 62), (2050, 12, 1026, 6)]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for k, (sw, sh, dw, dh) in enumerate(GEOM):
        fl = (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_AREA)[k % 4]
        r = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw + dh)
        if k == 0 and src in ("yuv420p", "nv12", "yuv420p10le") and dst in ("rgb565le", "x2rgb10le", "ayuv", "vuya", "xv30le", "y210le", "rgb48le", "gbrpf32le"):
            assert "two_pass" not in r[0] and "tile" not in r[0], (r[0], src, dst)
    run_case(854, 480, src, 1282, 720, dst, SWS_BICUBIC | BX, seed=3, device_frames=False)
