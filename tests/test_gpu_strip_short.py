"""-m gpu parity for the short-filter forms of the marching strip kernel (round 4): k_strip2.hip's register-staged instantiations
(sws_k_strip_short: rings of 3 / 4 / 6 row pairs, one staged chunk per lane, strips of 192 .. 320 luma / 64 .. 192 chroma columns) and the
LDS-DMA form for 8-bit planar sources (kernels_strip8.hpp, sws_k_strip_dma8: raw byte rows in LDS, byte pairs unpacked by v_perm in the
horizontal stage, windows starting at any byte, strips of up to 448 / 320 columns).  Reference arithmetic: hScale8To15_c (swscale.c:127-142),
yuv2planeX_8_c / yuv2plane1_8_c / yuv2planeX_10_c / yuv2nv12cX_c / yuv2p01x* (output.c:327-357, :468-589).
Every case is compared with the oracle; the same case with the family switched off (`no_strip_short`) must give the same bytes, since both
are compared with the same expectation."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SPLINE, SWS_FAST_BILINEAR)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
T0 = {"strip_min_w": 0}
FORMS = {
    "dma8": dict(T0),                                   # the planner's own choice: LDS-DMA form for planar sources on aligned frames
    "short": dict(T0, no_strip_dma8=1),                 # register-staged short instantiations
    "general": dict(T0, no_strip_short=1),              # round-3 kernels
}

SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "yuv411p", "yuv440p", "yuvj420p", "gray8", "nv12", "nv21", "nv16"]
DST = ["yuv420p", "yuv444p", "yuv422p", "yuv420p10le", "yuv444p12le", "yuv420p9le", "nv12", "nv21", "nv16", "p010le", "p012le", "gray8", "gray10le"]


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("sfmt", SRC)
def test_formats(form, sfmt):
    for dfmt in DST:
        if ("gray" in sfmt) != ("gray" in dfmt):
            continue
        for (sw, sh, dw, dh, fl) in ((644, 70, 322, 35, SWS_BILINEAR), (400, 66, 330, 54, SWS_BICUBIC)):
            run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=FORMS[form])


SCALERS = [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND]
# widths around the strip widths the planner picks (64 * cols columns, cols 1 .. 7) and ratios from 1:2 up to 3:1
GEOMS = [(1280, 72, 640, 36), (640, 36, 1280, 72), (1282, 50, 641, 25), (1278, 50, 639, 25), (900, 40, 449, 33), (896, 40, 447, 20), (960, 44, 320, 22), (963, 44, 321, 23),
         (770, 33, 385, 47), (512, 40, 383, 30), (1920, 30, 1280, 20), (1440, 28, 1920, 28), (1300, 31, 700, 31), (700, 64, 700, 32), (2600, 20, 1000, 10)]


@pytest.mark.parametrize("form", ["dma8", "short"])
@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: f"{g[0]}x{g[1]}to{g[2]}x{g[3]}")
def test_scalers_and_geometries(form, geom):
    sw, sh, dw, dh = geom
    for fl in SCALERS:
        for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv444p", "nv12"), ("yuv422p", "yuv420p10le"), ("nv12", "yuv420p"), ("nv21", "nv12"), ("nv24", "yuv444p")):
            run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw, tune=FORMS[form])


def test_chroma_positions_and_ranges():
    """odd window starts everywhere: shifted chroma sitings move the horizontal positions of the chroma planes (get_local_pos, utils.c:168-175)"""
    for hp in (-513, 0, 128, 256):
        for vp in (-513, 0, 256):
            opts = dict(dither=1, src_range=0, dst_range=0, src_h_chr_pos=hp, src_v_chr_pos=vp, dst_h_chr_pos=0, dst_v_chr_pos=128, threads=1)
            for form in ("dma8", "short"):
                run_case(1284, 60, "yuv420p", 642, 30, "yuv420p", SWS_BICUBIC | BX, seed=abs(hp + vp) + 1, opts=opts, tune=FORMS[form])
                run_case(700, 60, "yuv422p", 512, 44, "nv12", SWS_BILINEAR | BX, seed=abs(hp - vp) + 2, opts=opts, tune=FORMS[form])


def test_the_planner_names_the_kernel():
    from librempeg_amd import SwsContext
    for fmt, tune, want in (("yuv420p", {}, "sws_k_strip_dma8"), ("nv12", {}, "sws_k_strip_dma8"), ("yuv420p", {"no_strip_dma8": 1}, "sws_k_strip_short"),
                            ("yuv420p", {"no_strip_short": 1}, "sws_k_strip_march")):
        p = SwsContext(1280, 720, fmt, 640, 360, "yuv420p", SWS_BILINEAR | BX)
        for k, v in tune.items():
            p.set_option(k, v)
        assert p.path() == "main:strip_march"
        assert p.kernel_name() == want, (fmt, tune, p.kernel_name())
        p.close()


def test_full_size_frames():
    """C1 as BASELINE states it and the common ladder rungs, whole frames"""
    for form in ("dma8", "short"):
        assert run_case(1280, 720, "yuv420p", 640, 360, "yuv420p", SWS_BILINEAR | BX, seed=1, tune=FORMS[form])[0] == "main:strip_march"
        run_case(1920, 1080, "yuv420p", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2, tune=FORMS[form])
        run_case(3840, 2160, "yuv420p", 1920, 1080, "nv12", SWS_BICUBIC | BX, seed=3, tune=FORMS[form])
        run_case(3840, 2160, "nv12", 1920, 1080, "nv12", SWS_BICUBIC | BX, seed=5, tune=FORMS[form])
        run_case(1920, 1080, "nv12", 1280, 720, "yuv420p", SWS_BILINEAR | BX, seed=6, tune=FORMS[form])
        run_case(1920, 1080, "yuv444p", 1280, 720, "yuv420p10le", SWS_BILINEAR | BX, seed=4, tune=FORMS[form], device_frames=False)


def test_batches_unaligned_and_host_frames():
    """batches through one launch set; frames whose planes are not 16-byte aligned fall back to the register-staged forms (the DMA needs aligned rows)"""
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    from test_gpu_unaligned_frames import run_odd
    sw, sh, dw, dh, fl = 1280, 96, 640, 48, SWS_BILINEAR | BX
    for form in ("dma8", "short"):
        o = OL.Oracle(sw, sh, "yuv420p", dw, dh, "yuv420p", fl)
        p = SwsContext(sw, sh, "yuv420p", dw, dh, "yuv420p", fl)
        for k, v in FORMS[form].items():
            p.set_option(k, v)
        for n in (9, 2, 33):
            refs, srcs, dsts = [], [], []
            for k in range(n):
                s = OL.fill_random(OL.Frame("yuv420p", sw, sh), 500 + k + n)
                ref = OL.Frame("yuv420p", dw, dh, fill=0x21); assert o.scale(s, ref) == dh; refs.append(ref)
                hs = HostFrame("yuv420p", sw, sh)
                for a, b in zip(hs.planes, s.planes):
                    a[:] = b
                if k % 5 == 4:
                    hd = HostFrame("yuv420p", dw, dh)
                    for a in hd.planes:
                        a[:] = 0x21
                    srcs.append(hs); dsts.append(hd)
                else:
                    dd = DeviceFrame("yuv420p", dw, dh); dd.buf.fill_(0x21)
                    srcs.append(DeviceFrame("yuv420p", sw, sh).upload(hs)); dsts.append(dd)
            torch.cuda.synchronize()
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
            for k in range(n):
                out = dsts[k].download() if isinstance(dsts[k], DeviceFrame) else dsts[k]
                for pl, (a, b) in enumerate(zip(out.planes, refs[k].planes)):
                    rb = out.row_bytes[pl]
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (form, n, k, pl)
        p.close()
    for pad, shift, flip in ((0, 1, 0), (6, 3, 0), (13, 0, 1), (2, 2, 2)):
        run_odd(1284, 50, "yuv420p", 642, 26, "yuv420p", SWS_BICUBIC | BX, pad, shift, flip, nframes=2, tune=T0, seed=pad + 3)
        run_odd(644, 50, "yuv422p", 400, 40, "nv12", SWS_BILINEAR | BX, pad, shift, flip, nframes=3, tune=T0, seed=pad + 5)


@pytest.mark.parametrize("direct", [1, 0], ids=["direct", "split"])
@pytest.mark.parametrize("sfmt", ["nv12", "nv21", "nv16", "nv24", "nv42", "p010le", "p012le", "p010be", "p210le", "p410le", "p016le"])
def test_semi_planar_sources_into_packed_rgb(sfmt, direct):
    """decoder output -> display / inference: the strip-RGB kernels read a semi-planar source themselves on aligned frames (nv12 family: the LDS-DMA
    form's selectors de-interleave the chroma bytes, sws_k_strip_rgb8<..., NV>; p010 family: the 16-bit instantiation shifts the words down and
    de-interleaves them while staging, sws_k_strip_rgb<..., S16, P01X>) -- same bytes as through the split pass (no_striprgb_direct)"""
    tune = dict(T0, no_striprgb_direct=0 if direct else 1)
    for dfmt in ("rgb24", "bgra", "argb", "bgr24"):
        for (sw, sh, dw, dh, fl) in ((1280, 72, 640, 36, SWS_BICUBIC), (644, 70, 322, 35, SWS_BILINEAR), (640, 36, 1280, 72, SWS_BICUBIC), (1920, 30, 1280, 20, SWS_LANCZOS),
                                     (700, 64, 700, 32, SWS_BICUBIC | SWS_ACCURATE_RND), (1300, 31, 703, 31, SWS_AREA)):
            run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=tune)


def test_semi_planar_sources_into_packed_rgb_full_size_and_batches():
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    from test_gpu_unaligned_frames import run_odd
    assert run_case(3840, 2160, "nv12", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=8)[0] == "main:nvdirect+strip_rgb"
    assert run_case(3840, 2160, "p010le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=9)[0] == "main:nvdirect+strip_rgb"
    run_case(1920, 1080, "nv21", 1280, 720, "rgb24", SWS_BILINEAR | BX, seed=10, device_frames=False)
    for sfmt in ("nv12", "p010le"):
        sw, sh, dw, dh, fl = 1280, 96, 640, 48, SWS_BICUBIC | BX
        o = OL.Oracle(sw, sh, sfmt, dw, dh, "bgra", fl)
        p = SwsContext(sw, sh, sfmt, dw, dh, "bgra", fl)
        p.set_option("strip_min_w", 0)
        for n in (5, 2, 9):
            refs, srcs, dsts = [], [], []
            for k in range(n):
                s = OL.fill_random(OL.Frame(sfmt, sw, sh), 700 + k + n)
                ref = OL.Frame("bgra", dw, dh, fill=0x21); assert o.scale(s, ref) == dh; refs.append(ref)
                hs = HostFrame(sfmt, sw, sh)
                for a, b in zip(hs.planes, s.planes):
                    a[:] = b
                dd = DeviceFrame("bgra", dw, dh); dd.buf.fill_(0x21)
                srcs.append(DeviceFrame(sfmt, sw, sh).upload(hs)); dsts.append(dd)
            torch.cuda.synchronize()
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
            for k in range(n):
                out = dsts[k].download()
                assert np.array_equal(out.planes[0][:, :out.row_bytes[0]], refs[k].planes[0][:, :out.row_bytes[0]]), (sfmt, n, k)
        p.close()
        for pad, shift, flip in ((0, 1, 0), (6, 2, 0), (4, 0, 1)):       # unaligned / bottom-up frames: the split pass on aligned working copies
            run_odd(644, 50, sfmt, 400, 40, "rgb24", SWS_BICUBIC | BX, pad, shift, flip, nframes=2, tune=T0, seed=pad + 7)
