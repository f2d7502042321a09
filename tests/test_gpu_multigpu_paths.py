"""The N > 1 code paths of the library on the hardware that is there (one MI355X in the test box): the in-library sharding of
sws_scale_frames() with ndev = 1 and max_devices honoured, the host-frame staging path against per-frame sws_scale(), mixed resident /
host batches, cascaded contexts through the batch call, and bench.py's --inproc mode end to end.  A scaling curve needs an 8-GPU node:
none is measured here."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_BITEXACT, SWS_ACCURATE_RND

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BX = SWS_BITEXACT
CASES = [(320, 180, "yuv420p", 320, 180, "rgb24", SWS_BICUBIC | BX, {}), (320, 180, "nv12", 320, 180, "bgr0", SWS_BICUBIC | BX, {}),
         (256, 144, "yuv420p10le", 128, 72, "p010le", SWS_BILINEAR | BX, {}), (160, 90, "rgba", 96, 54, "yuv420p", SWS_BICUBIC | BX, dict(alpha_blend=1)),
         (160, 90, "yuv420p", 96, 54, "rgb24", SWS_BICUBIC | BX, dict(gamma_flag=1)), (160, 90, "yuv420p", 160, 90, "nv12", SWS_BICUBIC | BX, {})]


def _host(fmt, w, h, seed):
    src = OL.fill_random(OL.Frame(fmt, w, h), seed)
    hf = HostFrame(fmt, w, h)
    for a, b in zip(hf.planes, src.planes):
        a[:] = b
    return hf


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}" + ("-" + "-".join(c[7]) if c[7] else ""))
def test_batch_of_host_and_resident_frames_equals_single_calls(case):
    sw, sh, sf, dw, dh, df, flags, opts = case
    n = 7
    hsrc = [_host(sf, sw, sh, 50 + i) for i in range(n)]
    # reference results: one sws_scale() per frame on host buffers
    p1 = SwsContext(sw, sh, sf, dw, dh, df, flags, **opts)
    want = []
    for i in range(n):
        hd = HostFrame(df, dw, dh)
        assert p1.scale(hsrc[i], hd) >= 0
        want.append([a[:, :rb].copy() for a, rb in zip(hd.planes, hd.row_bytes)])
    p1.close()
    for max_devices in (0, 1):
        p = SwsContext(sw, sh, sf, dw, dh, df, flags, **opts)
        p.set_option("max_devices", max_devices)
        # frames 0, 2, 4, 6 live in HBM, frames 1, 3, 5 in host memory: resident frames go out as one launch set, host frames are staged
        srcs, dsts = [], []
        for i in range(n):
            if i % 2 == 0:
                srcs.append(DeviceFrame(sf, sw, sh).upload(hsrc[i])); dsts.append(DeviceFrame(df, dw, dh))
            else:
                srcs.append(hsrc[i]); dsts.append(HostFrame(df, dw, dh))
        torch.cuda.synchronize()
        assert p.scale_frames(srcs, dsts) == n
        p.sync()
        for i in range(n):
            out = dsts[i].download() if i % 2 == 0 else dsts[i]
            for a, b, rb in zip(out.planes, want[i], out.row_bytes):
                assert np.array_equal(a[:, :rb], b), (case, max_devices, i)
        p.close()


def test_bench_inproc_one_gpu():
    """bench.py --gpus 1 --inproc: one process, one context, one sws_scale_frames() call per step with the library doing the sharding"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--inproc", "--steps", "3", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "in-library" in line["config"]["sharding"]
