"""The product and the oracle refuse the same contexts: over the case generators of test_gpu_random.py (random format pairs,
sizes, scalers, flags; and the option-carrying construction), sws_getContext() / sws_init_context() of libswscale_hip.so
returns NULL / an error exactly when the oracle refuses the context.  Host-side only (no GPU needed): a case the GPU tests skip
because the oracle refuses it is a case the product refuses too."""
import pytest

import oracle_lib as OL
from librempeg_amd import SwsContext
from test_gpu_random import _cases, _opt_cases


def _refuses(make):
    try:
        c = make()
    except Exception:
        return True
    if hasattr(c, "close"):
        c.close()
    return False


def test_refusals_agree_plain():
    bad = []
    for sw, sh, sf, dw, dh, df, flags, k in _cases(6000, 20260928):
        o = _refuses(lambda: OL.Oracle(sw, sh, sf, dw, dh, df, flags))
        p = _refuses(lambda: SwsContext(sw, sh, sf, dw, dh, df, flags))
        if o != p:
            bad.append((k, sf, sw, sh, df, dw, dh, hex(flags), "oracle refuses" if o else "product refuses"))
    assert not bad, bad[:10]


def test_refusals_agree_with_options():
    bad = []
    for sw, sh, sf, dw, dh, df, flags, k, opts, cs in _opt_cases(3000, 777):
        o = _refuses(lambda: OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts))
        p = _refuses(lambda: SwsContext(sw, sh, sf, dw, dh, df, flags, **opts))
        if o != p:
            bad.append((k, sf, sw, sh, df, dw, dh, hex(flags), opts, "oracle refuses" if o else "product refuses"))
    assert not bad, bad[:10]
