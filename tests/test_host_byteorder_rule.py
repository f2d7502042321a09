"""CPU box: the product's plan for "the same format in the other byte order" (csrc/context.cpp choose_unscaled; the reference's bswap_16bpc / bswap_32bpc rules and the simple-copy
rule, swscale_unscaled.c:545-597, :2560-2668).  Found by tools/ref/ref_crosscheck.py in round 6: with SWS_SRC_V_CHR_DROP the copy rule no longer applies to planar YUV, bswap_16bpc
writes srcSliceH >> chrDstVSubSample rows of EVERY plane (the oracle restates that), YUVA and semi-planar formats go through the scaler.  The product takes its copy plans where all rows
are swapped, the scaler where the reference does, and refuses the half-written luma plane."""
import ctypes as C

import pytest

from librempeg_amd import SwsContext, SWS_BICUBIC, SWS_BITEXACT
import oracle_lib as OL

CASES = [  # (src, dst, drop) -> product path, or None = refused
    ("yuv420p10be", "yuv420p10le", 0, "unscaled:planarCopy"), ("yuv420p10be", "yuv420p10le", 1, None), ("yuv440p10le", "yuv440p10be", 2, None),
    ("yuv444p10be", "yuv444p10le", 1, "unscaled:planarCopy"), ("yuv422p10be", "yuv422p10le", 1, "unscaled:planarCopy"),
    ("yuva420p10be", "yuva420p10le", 0, "unscaled:planarCopy"), ("yuva420p10be", "yuva420p10le", 1, "main:"), ("p010be", "p010le", 1, "main:"),
    ("gbrp10be", "gbrp10le", 1, "unscaled:planarCopy"), ("rgb565be", "rgb565le", 1, "unscaled:packedCopy"), ("gbrpf32be", "gbrpf32le", 1, "unscaled:planarCopy"),
    ("gray16be", "gray16le", 1, "unscaled:planarCopy"), ("yuv420p10le", "yuv420p10le", 1, "unscaled:planarCopy"),
]


@pytest.mark.parametrize("sf,df,drop,want", CASES)
def test_byte_order_rule(sf, df, drop, want):
    flags = SWS_BICUBIC | SWS_BITEXACT | (drop << 16)
    OL.Oracle(64, 36, sf, 64, 36, df, flags)                  # the oracle (held to the reference's answers: tests/test_oracle_reference_answers_r06.py) takes them all
    if want is None:
        with pytest.raises(Exception):
            SwsContext(64, 36, sf, 64, 36, df, flags)
        return
    p = SwsContext(64, 36, sf, 64, 36, df, flags)
    p.set_option("dry_plan", 1)
    p.L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert p.L.sws_hip_plan(p.c, (C.c_uint64 * 3)()) == 0
    assert p.path().startswith(want), p.path()
    p.close()
