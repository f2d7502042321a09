"""fate-sws-floatimg-cmp through the oracle: pins the float RGB reader (C5's reader) and the float planar writer against the
reference's own known answers (tests/golden/fate_sws_floatimg_cmp.txt)."""
import numpy as np
import pytest

import floatimg_cmp as FC
import oracle_lib as OL


def roundtrip(make_ctx, Frame, fmt):
    w, h = FC.W, FC.H
    src = Frame("gbrpf32le", w, h)
    for pl, v in zip(src.planes, FC.source_planes()):
        pl[:, :4 * w] = v.view(np.uint8).reshape(h, 4 * w)
    mid = Frame(fmt, w, h)
    back = Frame("gbrpf32le", w, h)
    assert make_ctx("gbrpf32le", fmt).scale(src, mid) == h
    assert make_ctx(fmt, "gbrpf32le").scale(mid, back) == h
    outs = [pl[:, :4 * w].copy().view(np.float32).reshape(h, w) for pl in back.planes]
    return FC.stats(FC.source_planes(), outs)


@pytest.mark.parametrize("row", FC.golden(), ids=lambda r: r[0])
def test_floatimg_cmp_oracle(row):
    fmt, avg, mn, mx = row
    if fmt not in OL.FMT:
        pytest.skip(f"{fmt} is not built yet (DESIGN.md 7)")
    got = roundtrip(lambda s, d: OL.Oracle(FC.W, FC.H, s, FC.W, FC.H, d, FC.SWS_BILINEAR), OL.Frame, fmt)
    assert got == (avg, mn, mx)
