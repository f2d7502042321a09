"""-m gpu: a conversion's result may depend only on the bytes of the caller's planes, and may touch only the destination's.

Every rare parity event of rounds 1 - 5 happened with several test processes sharing the GPU and was never reproduced alone -- what a kernel that
READS a few bytes outside a plane would look like: alone, the allocator hands every run the same neighbours; under xdist the neighbours change
from run to run.  This test makes the neighbours the variable: source and destination live inside larger allocations whose surroundings
(GUARD bytes before the first plane and behind the last one) are filled with 0x00, then with 0xFF, then with a ramp; the destination -- every byte
of its allocation, row padding included -- must come out the same each time, and the destination's own guard bands must keep their fill.
No oracle is involved (the parity tests compare results; this one compares the product with itself), so the draws can be large:
the generators are those of tests/test_gpu_random.py."""
import os

import numpy as np
import pytest

import oracle_lib as OL
import test_gpu_random as R
from librempeg_amd.swscale import DeviceFrame, HostFrame, SwsContext, image_layout

pytestmark = pytest.mark.gpu
GUARD = int(os.environ.get("SWS_GUARD_BYTES", "8192"))      # (hunts for far writes: 1 MiB and more)


class GuardedFrame(DeviceFrame):
    """DeviceFrame inside a larger allocation: GUARD bytes of surroundings on both sides of the picture"""

    def __init__(self, fmt, w, h, device="cuda:0"):
        import torch
        if os.environ.get("SWS_SUITE_ON_EMU") == "1":       # (tests/conftest.py: the suite against the x86 emulation build, CPU box)
            device = "cpu"
        self.fmt, self.w, self.h = fmt, w, h
        self.linesize, self.offset, self.total = image_layout(fmt, w, h, 256)
        self.buf = torch.zeros(GUARD + self.total + 256 + GUARD, dtype=torch.uint8, device=device)
        base = self.buf.data_ptr() + GUARD
        self.base = (base + 255) // 256 * 256
        self._shift = self.base - self.buf.data_ptr()
        from librempeg_amd.swscale import plane_layout
        self.nplanes = len(plane_layout(fmt, w, h))
        self.row_bytes = [rb for rb, _ in plane_layout(fmt, w, h)]
        self.rows = [r for _, r in plane_layout(fmt, w, h)]

    def fill_guards(self, mode):
        import torch
        lo, hi = self.buf[:self._shift], self.buf[self._shift + self.total:]
        for g in (lo, hi):
            if mode == 2:
                g.copy_((torch.arange(g.numel(), device=g.device) * 37 + 11).to(torch.uint8))
            else:
                g.fill_(0xFF if mode else 0)

    def guards(self):
        return self.buf[:self._shift].cpu().numpy().copy(), self.buf[self._shift + self.total:].cpu().numpy().copy()

    def inside(self):
        return self.buf[self._shift:self._shift + self.total].cpu().numpy().copy()


def _run(sw, sh, sf, dw, dh, df, flags, seed, opts=None, cs=None, tune=None):
    import torch
    src = OL.fill_random(OL.Frame(sf, sw, sh), seed)
    hs = HostFrame(sf, sw, sh)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    ds = GuardedFrame(sf, sw, sh).upload(hs)
    dd = GuardedFrame(df, dw, dh)
    outs = []
    path = None
    for mode in (0, 1, 2):
        # (a context per fill: the error-diffusion writers carry their error line from one conversion of a context to the next, swscale.c:1084-1086)
        try:
            p = SwsContext(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
        except RuntimeError:
            pytest.skip("refused")
        for k, v in (tune or {}).items():
            p.set_option(k, v)
        if cs and p.set_colorspace(*cs) < 0:
            p.close()
            pytest.skip("colourspace details refused")
        ds.fill_guards(mode); dd.fill_guards(mode)
        dd.buf[dd._shift:dd._shift + dd.total].fill_(0x5A)
        g0 = dd.guards(); s0 = ds.guards()
        torch.cuda.synchronize()
        ret = p.scale(ds, dd)
        p.sync()
        path = p.path()
        p.close()
        if ret < 0:
            pytest.skip(f"sws_scale refused: {ret}")
        g1 = dd.guards(); s1 = ds.guards()
        for a, b in zip(g0 + s0, g1 + s1):
            assert np.array_equal(a, b), f"{sf}->{df} {sw}x{sh}->{dw}x{dh} flags={flags:#x} path={path}: bytes OUTSIDE the pictures were written (guard mode {mode})"
        outs.append(dd.inside())
    for mode in (1, 2):
        if not np.array_equal(outs[0], outs[mode]):
            bad = np.flatnonzero(outs[0] != outs[mode])
            raise AssertionError(f"{sf}->{df} {sw}x{sh}->{dw}x{dh} flags={flags:#x} path={path}: the result depends on bytes outside the caller's planes "
                                 f"({len(bad)} bytes differ between guard fills 0 and {mode}, first at offset {bad[0]} of the destination allocation; plane offsets {dd.offset}, "
                                 f"linesizes {dd.linesize})")


_N, _SEED = os.environ.get("SWS_RANDOM_N"), os.environ.get("SWS_RANDOM_SEED")
_ID = lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}"


@pytest.mark.parametrize("case", R._cases(int(_N or 1500), int(_SEED or 616)), ids=_ID)
def test_guard_bands_format_matrix(case):
    sw, sh, sf, dw, dh, df, flags, k = case
    _run(sw, sh, sf, dw, dh, df, flags, k + 1)


@pytest.mark.parametrize("case", R._strip_cases(int(_N or 1500), int(_SEED or 617)), ids=_ID)
def test_guard_bands_strip_family(case):
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune = case
    _run(sw, sh, sf, dw, dh, df, flags, k + 3, opts or None, cs, tune)


@pytest.mark.parametrize("case", R._strip_cases(int(_N or 1500), int(_SEED or 618), R.R4_SRC, R.R4_DST), ids=_ID)
def test_guard_bands_round4_routes(case):
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune = case
    _run(sw, sh, sf, dw, dh, df, flags, k + 5, opts or None, cs, tune)
