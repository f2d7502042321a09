"""-m gpu: sws_scale_frames() batches whose per-frame working pictures exceed the helper passes' budget (Tuning::work_mb, dev_exec.hip
launch_plan_le) are cut into sub-batches that reuse the working buffers.  With the budget forced down to 1 MiB every family of helper pass
(reader pre-pass, 4:2:2 / semi-planar splits, 4:2:2 join, full-chroma sums, alpha launches, staging of unaligned frames) runs a 7-frame call
as several sub-batches; results must equal the oracle frame by frame, and a second call with other frames must follow."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT

CASES = [
    # src, dst, sw, sh, dw, dh, flags, expected path fragment
    ("rgb24", "yuv420p", 644, 70, 516, 56, SWS_BICUBIC | BX, "rgbread"),
    ("bgra", "rgb24", 640, 96, 320, 48, SWS_BICUBIC | BX, "fullchr_rgb"),
    ("bgra", "bgra", 640, 96, 400, 60, SWS_BILINEAR | BX, "fullchr_rgb"),
    ("yuyv422", "yuv420p", 640, 64, 426, 42, SWS_BICUBIC | BX, "split422"),
    ("nv12", "bgra", 640, 64, 320, 32, SWS_BICUBIC | BX, "splitnv"),
    ("yuv420p", "uyvy422", 640, 64, 320, 32, SWS_BICUBIC | BX, "join422"),
    ("yuva420p", "bgra", 640, 64, 320, 32, SWS_BICUBIC | BX, "alpha"),
    ("p010le", "bgra", 640, 64, 320, 32, SWS_LANCZOS | BX, "strip_rgb"),
    ("yuv420p", "gbrp", 640, 64, 320, 32, SWS_BICUBIC | BX, "fullchr"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
@pytest.mark.parametrize("work_mb", [1, 3])
def test_sub_batches(case, work_mb):
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    src, dst, sw, sh, dw, dh, flags, frag = case
    o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
    p = SwsContext(sw, sh, src, dw, dh, dst, flags)
    p.set_option("strip_min_w", 0)
    p.set_option("work_mb", work_mb)
    if frag == "fullchr_rgb":
        p.set_option("no_strip_rgb2rgb", 1)        # (likewise: the helper passes the one-launch RGB -> RGB form replaces)
    if frag in ("rgbread", "split422"):
        p.set_option("no_strip_rgbsrc", 1)         # (likewise: the reader pre-pass the one-launch form replaces)
    if frag == "splitnv":
        p.set_option("no_striprgb_direct", 1)      # (this file is about the helper passes: keep the split pass the direct reader of round 4 replaces)
    seed = 900
    for n in (7, 3, 7):
        refs, srcs, dsts = [], [], []
        for k in range(n):
            seed += 1
            s = OL.fill_random(OL.Frame(src, sw, sh), seed)
            ref = OL.Frame(dst, dw, dh, fill=0x5C)
            assert o.scale(s, ref) >= 0
            refs.append(ref)
            hs = HostFrame(src, sw, sh)
            for a, b in zip(hs.planes, s.planes):
                a[:] = b
            dd = DeviceFrame(dst, dw, dh); dd.buf.fill_(0x5C)
            srcs.append(DeviceFrame(src, sw, sh).upload(hs)); dsts.append(dd)
        torch.cuda.synchronize()
        assert p.scale_frames(srcs, dsts) == n
        p.sync()
        assert frag in p.path(), p.path()
        for k in range(n):
            out = dsts[k].download()
            for pl, (a, b) in enumerate(zip(out.planes, refs[k].planes)):
                rb = out.row_bytes[pl]
                assert np.array_equal(a[:, :rb], b[:, :rb]), (case, work_mb, n, k, pl, p.path())
    p.close()


def test_timed_region_spans_the_sub_batches():
    """sws_hip_last_kernel_ms() of a call that went out as sub-batches covers all of them (event 0 before the first, event 1 after the last)"""
    import torch
    from librempeg_amd import SwsContext, DeviceFrame
    sw, sh, dw, dh, n = 1280, 720, 640, 360, 12
    times = []
    for work_mb in (2048, 4):
        p = SwsContext(sw, sh, "bgra", dw, dh, "rgb24", SWS_BICUBIC | BX)
        p.set_option("work_mb", work_mb)
        p.set_option("no_strip_rgb2rgb", 1)        # (the helper-pass form: the one-launch form has no working pictures to cut a batch for)
        p.set_timing(True)
        srcs = [DeviceFrame("bgra", sw, sh) for _ in range(n)]; dsts = [DeviceFrame("rgb24", dw, dh) for _ in range(n)]
        torch.cuda.synchronize()
        for _ in range(3):
            assert p.scale_frames(srcs, dsts) == n
            p.sync()
        times.append(p.last_kernel_ms())
        p.close()
    assert times[0] > 0 and times[1] > 0
    assert times[1] > 0.5 * times[0], times      # (one sub-batch of twelve would be ~1/12 of the call)
