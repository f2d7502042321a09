"""-m gpu: the frame tables of batched launches live in a ring (TableRing, dev_state.hip table_upload / table_batch_end): a call with new frame
pointers takes the next span instead of waiting for the stream, spans are recycled behind an event recorded after the launch set that read them.
Many sws_scale_frames() calls are queued back to back on DIFFERENT frame sets with no synchronisation in between -- enough of them to wrap the
ring several times and, with larger batches, to regrow it -- and every output is compared with the oracle afterwards."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_BITEXACT

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT

CASES = [
    # src, dst, sw, sh, dw, dh, flags, options: one-launch forms and helper-pass forms (several tables per call)
    ("yuv420p", "rgb24", 128, 32, 128, 32, SWS_BICUBIC | BX, {}),
    ("yuv420p", "yuv420p", 640, 48, 320, 24, SWS_BILINEAR | BX, {"strip_min_w": 0}),
    ("bgra", "rgb24", 640, 48, 320, 24, SWS_BICUBIC | BX, {"strip_min_w": 0, "no_strip_rgb2rgb": 1}),
    ("rgb24", "yuv420p", 644, 40, 516, 32, SWS_BICUBIC | BX, {"strip_min_w": 0, "no_strip_rgbsrc": 1, "work_mb": 1}),
    ("yuva420p", "bgra", 640, 32, 320, 16, SWS_BICUBIC | BX, {"strip_min_w": 0}),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_calls_in_flight_on_distinct_frame_sets(case):
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    src, dst, sw, sh, dw, dh, flags, opts = case
    o = OL.Oracle(sw, sh, src, dw, dh, dst, flags)
    p = SwsContext(sw, sh, src, dw, dh, dst, flags)
    for k, v in opts.items():
        p.set_option(k, v)
    # a pool of distinct source pictures and their oracle answers
    pool = []
    for k in range(6):
        s = OL.fill_random(OL.Frame(src, sw, sh), 4100 + k)
        ref = OL.Frame(dst, dw, dh, fill=0x5C)
        assert o.scale(s, ref) >= 0
        hs = HostFrame(src, sw, sh)
        for a, b in zip(hs.planes, s.planes):
            a[:] = b
        pool.append((DeviceFrame(src, sw, sh).upload(hs), ref))
    torch.cuda.synchronize()
    # 3 rounds x 40 calls of 5 .. 29 frames (a ring of 512 entries wraps every few calls; the 200-frame call regrows it), nothing waited for inside a round
    rng = np.random.RandomState(7)
    for rnd in range(3):
        calls = []
        for i in range(40):
            n = 200 if (rnd == 1 and i == 17) else int(rng.randint(5, 30))
            idx = [int(rng.randint(0, len(pool))) for _ in range(n)]
            dsts = [DeviceFrame(dst, dw, dh) for _ in range(n)]
            for dd in dsts:
                dd.buf.fill_(0x5C)
            calls.append((idx, dsts))
        torch.cuda.synchronize()
        for idx, dsts in calls:
            assert p.scale_frames([pool[j][0] for j in idx], dsts) == len(idx)
        p.sync()
        for ci, (idx, dsts) in enumerate(calls):
            for k, j in enumerate(idx):
                out = dsts[k].download()
                for pl, (a, b) in enumerate(zip(out.planes, pool[j][1].planes)):
                    rb = out.row_bytes[pl]
                    assert np.array_equal(a[:, :rb], b[:, :rb]), (case[:2], rnd, ci, k, pl, p.path())
    p.close()


def test_identical_table_is_not_uploaded_again_and_changes_are_seen():
    """the same frame set twice (cache hit), then the same sources into other destinations: the second table must reach the kernels"""
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    sw, sh = 256, 32
    o = OL.Oracle(sw, sh, "yuv420p", sw, sh, "rgb24", SWS_BICUBIC | BX)
    p = SwsContext(sw, sh, "yuv420p", sw, sh, "rgb24", SWS_BICUBIC | BX)
    srcs, refs = [], []
    for k in range(4):
        s = OL.fill_random(OL.Frame("yuv420p", sw, sh), 77 + k)
        ref = OL.Frame("rgb24", sw, sh, fill=0)
        assert o.scale(s, ref) >= 0
        hs = HostFrame("yuv420p", sw, sh)
        for a, b in zip(hs.planes, s.planes):
            a[:] = b
        srcs.append(DeviceFrame("yuv420p", sw, sh).upload(hs)); refs.append(ref)
    d1 = [DeviceFrame("rgb24", sw, sh) for _ in range(4)]
    d2 = [DeviceFrame("rgb24", sw, sh) for _ in range(4)]
    for d in d1 + d2:
        d.buf.fill_(0)
    torch.cuda.synchronize()
    assert p.scale_frames(srcs, d1) == 4
    assert p.scale_frames(srcs, d1) == 4
    assert p.scale_frames(srcs, d2) == 4
    assert p.scale_frames(srcs[::-1], d1) == 4
    p.sync()
    for k in range(4):
        a = d2[k].download(); b = d1[k].download()
        rb = a.row_bytes[0]
        assert np.array_equal(a.planes[0][:, :rb], refs[k].planes[0][:, :rb])
        assert np.array_equal(b.planes[0][:, :rb], refs[3 - k].planes[0][:, :rb])
    p.close()
