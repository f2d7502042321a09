"""The streaming form of the layout / depth converters (kernels_layout.hpp: planarCopyWrapper incl. DITHER_COPY, planarToNv12 / Nv24,
nv12 / nv24ToPlanar, yuyv / uyvy <-> planar) against the oracle: widths around the 16-byte chunk and the 1 KiB wave boundaries, odd sizes,
slices, every dither mode of the depth reduction, both chunk counts per lane; and against the element-per-thread kernels it replaces."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT, SWS_ACCURATE_RND
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT

PAIRS = [("yuv420p", "nv12"), ("yuv420p", "nv21"), ("nv12", "yuv420p"), ("nv21", "yuv420p"), ("yuv444p", "nv24"), ("yuv444p", "nv42"), ("nv24", "yuv444p"),
         ("nv42", "yuv444p"), ("yuv420p", "yuv420p10le"), ("yuv422p", "yuv422p16le"), ("yuv444p", "yuv444p9le"), ("yuvj420p", "yuv420p12le"),
         ("yuv420p10le", "yuv420p"), ("yuv444p16le", "yuv444p"), ("yuv422p12le", "yuv422p10le"), ("yuv420p9le", "yuv420p14le"), ("yuv420p10le", "yuv420p16le"),
         ("yuv420p", "yuv420p"), ("yuv420p10le", "yuv420p10le"), ("gray8", "yuv420p"), ("gray10le", "yuv444p12le"), ("yuv420p", "gray8"), ("yuva420p", "yuva420p10le"),
         ("yuv420p", "yuva420p"), ("yuva444p10le", "yuva444p"), ("nv12", "p010le"), ("p010le", "nv12"), ("yuv420p10be", "yuv420p"), ("yuv444p10msble", "yuv444p"),
         ("yuyv422", "yuv420p"), ("uyvy422", "yuv420p"), ("yuyv422", "yuv422p"), ("uyvy422", "yuv422p"), ("yvyu422", "yuv420p"), ("yvyu422", "yuv422p"),
         ("yuv422p", "yuyv422"), ("yuv422p", "uyvy422"), ("yuv420p", "yuyv422"), ("yuv420p", "uyvy422")]
SIZES = [(1, 2), (2, 2), (7, 3), (15, 5), (16, 4), (17, 7), (31, 2), (33, 9), (63, 6), (64, 8), (65, 3), (127, 5), (129, 4), (255, 2), (257, 6), (511, 3), (513, 2), (1023, 4),
         (1024, 2), (1025, 3), (2047, 2), (2049, 5), (4094, 2), (4098, 3)]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_layout_sizes(pair):
    sf, df = pair
    for i, (w, h) in enumerate(SIZES):
        try:
            OL.Oracle(w, h, sf, w, h, df, SWS_BICUBIC | BX)
        except RuntimeError:
            continue
        run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=i)
        if i % 4 == 0:
            run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=i + 1, tune=dict(layout_ch=2))
            run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=i + 2, device_frames=False)


@pytest.mark.parametrize("dither", [0, 1, 2, 3, 4, 5], ids=["none", "auto", "bayer", "ed", "a_dither", "x_dither"])
@pytest.mark.parametrize("pair", [("yuv420p10le", "yuv420p"), ("yuv444p16le", "yuv444p"), ("yuv422p12le", "yuv422p10le"), ("yuv444p14le", "yuv444p9le"),
                                  ("gray16le", "gray8"), ("yuva444p16le", "yuva444p")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_dither_copy_modes(pair, dither):
    sf, df = pair
    for (w, h) in ((97, 19), (64, 16), (1031, 9)):
        for rng in (dict(), dict(src_range=1, dst_range=1)):
            run_case(w, h, sf, w, h, df, SWS_BICUBIC | BX, seed=w, opts=dict(dither=dither, **rng))


def test_stream_equals_element_kernels():
    """the same pictures through the streaming kernel and through the element-per-thread kernels it replaces (no_layout_stream)"""
    for sf, df in PAIRS:
        for (w, h) in ((130, 10), (1030, 6)):
            try:
                OL.Oracle(w, h, sf, w, h, df, SWS_BICUBIC | BX)
            except RuntimeError:
                continue
            src = OL.fill_random(OL.Frame(sf, w, h), 21)
            outs = []
            for off in (0, 1):
                p = SwsContext(w, h, sf, w, h, df, SWS_BICUBIC | BX)
                p.set_option("no_layout_stream", off)
                hs = HostFrame(sf, w, h)
                for a, b in zip(hs.planes, src.planes):
                    a[:] = b
                hd = HostFrame(df, w, h)
                for a in hd.planes:
                    a[:] = 0x5A
                assert p.scale(hs, hd) >= 0
                outs.append([a[:, :rb].copy() for a, rb in zip(hd.planes, hd.row_bytes)])
            assert all(np.array_equal(a, b) for a, b in zip(*outs)), (sf, df, w, h)
