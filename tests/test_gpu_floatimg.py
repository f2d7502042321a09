"""fate-sws-floatimg-cmp through the HIP library (C-ABI): the product reproduces the reference's known answers for the float RGB
reader and writer (tests/golden/fate_sws_floatimg_cmp.txt), from host frames and from HBM."""
import numpy as np
import pytest

import floatimg_cmp as FC
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, PIX_FMT

pytestmark = pytest.mark.gpu


def roundtrip(fmt, device):
    w, h = FC.W, FC.H
    src = HostFrame("gbrpf32le", w, h)
    for pl, v in zip(src.planes, FC.source_planes()):
        pl[:, :4 * w] = v.view(np.uint8).reshape(h, 4 * w)
    back = HostFrame("gbrpf32le", w, h)
    c0 = SwsContext(w, h, "gbrpf32le", w, h, fmt, FC.SWS_BILINEAR)
    c1 = SwsContext(w, h, fmt, w, h, "gbrpf32le", FC.SWS_BILINEAR)
    if device:
        import torch
        dsrc, dmid, dback = DeviceFrame("gbrpf32le", w, h).upload(src), DeviceFrame(fmt, w, h), DeviceFrame("gbrpf32le", w, h)
        torch.cuda.synchronize()
        assert c0.scale(dsrc, dmid) == h
        c0.sync()
        assert c1.scale(dmid, dback) == h
        c1.sync()
        back = dback.download(back)
    else:
        mid = HostFrame(fmt, w, h)
        assert c0.scale(src, mid) == h
        assert c1.scale(mid, back) == h
    outs = [pl[:, :4 * w].copy().view(np.float32).reshape(h, w) for pl in back.planes]
    return FC.stats(FC.source_planes(), outs)


@pytest.mark.parametrize("device", [False, True], ids=["host", "hbm"])
@pytest.mark.parametrize("row", FC.golden(), ids=lambda r: r[0])
def test_floatimg_cmp_hip(row, device):
    fmt, avg, mn, mx = row
    if fmt not in PIX_FMT:
        pytest.skip(f"{fmt} is not built yet (DESIGN.md 7)")
    assert roundtrip(fmt, device) == (avg, mn, mx)
