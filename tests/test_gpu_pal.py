"""Palette-expanded sources (usePal, swscale_internal.h:937-950): pal8 with its 256-word palette in data[1], and rgb8 / bgr8 /
rgb4_byte / bgr4_byte whose palette ff_update_palette builds from the bit fields (swscale.c:873-951).  The scaler reads them through
palToY_c / palToUV_c / palToA_c (input.c:474-512); unscaled conversions to the byte RGB formats and gbrp / gbrap go through
palToRgbWrapper / palToGbrpWrapper (swscale_unscaled.c:600-683, :2619-2630)."""
import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_FAST_BILINEAR, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT,
                           SwsContext, HostFrame, DeviceFrame)
from test_gpu_parity import run_case, PAL_IN

BX = SWS_BITEXACT
BYTE_RGB = ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "gbrp", "gbrap"]
DSTS = ["yuv420p", "yuva420p", "yuv444p10le", "rgb565le", "rgb48le", "gray8", "ya8", "nv12", "gbrp12le", "rgb8", "bgr4", "monob", "grayf32le", "yuva444p16be"]


@pytest.mark.gpu
@pytest.mark.parametrize("src", PAL_IN)
@pytest.mark.parametrize("dst", BYTE_RGB)
def test_palette_wrappers(src, dst):
    for w, h in ((64, 48), (61, 37), (1, 1), (200, 3)):
        path, opath = run_case(w, h, src, w, h, dst, SWS_BICUBIC | BX, seed=w)
        assert (path, opath) == ("unscaled:palToRgb", "palToRgb")
        run_case(w, h, src, w, h, dst, SWS_POINT, seed=w + 1, device_frames=False)
    run_case(64, 48, src, 40, 30, dst, SWS_BICUBIC | BX, seed=5)      # scaled: the readers
    run_case(64, 48, src, 80, 50, dst, SWS_BILINEAR | BX | SWS_FULL_CHR_H_INT, seed=6, device_frames=False)


@pytest.mark.gpu
@pytest.mark.parametrize("src", PAL_IN)
@pytest.mark.parametrize("dst", DSTS)
def test_palette_readers(src, dst):
    run_case(96, 64, src, 60, 40, dst, SWS_BICUBIC | BX, seed=1)
    run_case(97, 63, src, 97, 63, dst, SWS_LANCZOS | BX | SWS_ACCURATE_RND, seed=2)   # same size, no wrapper for these destinations (rgb8 -> rgb8 is a copy)
    run_case(64, 40, src, 128, 80, dst, SWS_FAST_BILINEAR | BX, seed=3, device_frames=False)


@pytest.mark.gpu
def test_every_frame_of_a_batch_has_its_own_palette():
    sw, sh, dw, dh, n = 64, 48, 40, 30, 5
    for dst, flags in (("rgba", SWS_BICUBIC | BX), ("yuva420p", SWS_BILINEAR | BX)):
        for dwh in ((sw, sh), (dw, dh)):
            o = OL.Oracle(sw, sh, "pal8", dwh[0], dwh[1], dst, flags)
            p = SwsContext(sw, sh, "pal8", dwh[0], dwh[1], dst, flags)
            refs, srcs, dsts = [], [], []
            for k in range(n):
                src = OL.fill_random(OL.Frame("pal8", sw, sh), 40 + k)
                ref = OL.Frame(dst, *dwh)
                assert o.scale(src, ref) == dwh[1]
                refs.append(ref)
                hs = HostFrame("pal8", sw, sh)
                for a, b in zip(hs.planes, src.planes):
                    a[:] = b
                srcs.append(DeviceFrame("pal8", sw, sh).upload(hs))
                dsts.append(DeviceFrame(dst, *dwh))
            import torch
            torch.cuda.synchronize()
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
            for k in range(n):
                out = dsts[k].download(HostFrame(dst, *dwh))
                for a, b, rb in zip(out.planes, refs[k].planes, out.row_bytes):
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (dst, dwh, k)
            p.close()


def test_palette_formats_in_the_format_queries(hiplib):
    L = hiplib
    assert L.sws_isSupportedInput(LA.PIX_FMT["pal8"]) == 1 and L.sws_isSupportedOutput(LA.PIX_FMT["pal8"]) == 0
    for f in ("rgb8", "bgr8", "rgb4_byte", "bgr4_byte"):
        assert L.sws_isSupportedInput(LA.PIX_FMT[f]) == 1 and L.sws_isSupportedOutput(LA.PIX_FMT[f]) == 1
    for f in ("rgb4", "bgr4"):
        assert L.sws_isSupportedInput(LA.PIX_FMT[f]) == 0 and L.sws_isSupportedOutput(LA.PIX_FMT[f]) == 1
    for make in (OL.Oracle, SwsContext):
        with pytest.raises(RuntimeError):
            make(64, 48, "yuv420p", 64, 48, "pal8", SWS_BICUBIC | BX)
        with pytest.raises(RuntimeError):
            make(64, 48, "rgb4", 64, 48, "yuv420p", SWS_BICUBIC | BX)
