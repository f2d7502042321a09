"""The oracle against answers of the REAL reference (tests/golden/reference_answers_r06.json, made by tools/gen_crosscheck_golden.py from a C-only reference build) for the two
divergences round 6's cross-check found and fixed in oracle and product: the 8 / 4 bpp ordered-dither converters' last tail pair (yuv2rgb.c:283-318), and yuva420p10le /
yuva420p16le into p010le / p016le at the same size (planarToP01xWrapper, swscale_unscaled.c:2432-2439).  The GPU suite compares the product with this oracle on the same paths."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as OL

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_answers_r06.json")))


@pytest.mark.parametrize("g", G["cases"], ids=lambda g: "{2}_{0}x{1}-{5}".format(*g["case"]))
def test_oracle_equals_the_reference(g):
    sw, sh, sf, dw, dh, df, flags = g["case"]
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **g.get("opts", {}))
    src = OL.fill_random(OL.Frame(sf, sw, sh), g["seed"])
    dst = OL.Frame(df, dw, dh, fill=g["prefill"])
    assert o.scale(src, dst) == dh
    data = b"".join(np.ascontiguousarray(a[:, :rb]).tobytes() for a, rb in zip(dst.planes, dst.row_bytes))
    assert hashlib.md5(data).hexdigest() == g["md5"]
