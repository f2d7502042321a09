"""Stress loop for the one unreproduced failure of round 3: test_random_batches[9782-nv21_280x50-nv21_225x45-80200] --
the strip kernel with the semi-planar reader and writer, no helper pass, three sws_scale_frames() calls of 4 / 2 / 3 frames on ONE
context, the last mixing host frames in -- run by several processes on one GPU.

    python tools/stress_batches.py SECONDS [TAG] [MODE]

MODE: "exact" (default) loops the failing case only; "near" also draws neighbours (other semi-planar / planar pairs, sizes around it,
other batch sizes).  Every mismatch is reported with what the wrong bytes look like: still the 0x33 fill (a band that was not written),
equal to another frame's expectation of the same call (a frame-table mix-up), or neither (arithmetic on wrong inputs).
Options through the environment: SWS_STRESS_TUNE="no_strip_fuse=1,..." sets sws_hip_set_option()s on every context."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame

secs = float(sys.argv[1]); tag = sys.argv[2] if len(sys.argv) > 2 else "0"; mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
tune_env = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in os.environ.get("SWS_STRESS_TUNE", "").split(",") if kv)
BASE = (280, 50, "nv21", 225, 45, "nv21", 0x80200, 9782, {"strip_min_w": 0}, [4, 2, 3])
rng = random.Random(hash(tag) & 0xFFFF)


def draw():
    if mode == "exact" or rng.random() < 0.3:
        return BASE
    sf, df = rng.choice(["nv21", "nv12", "yuv420p", "nv16", "yuv422p"]), rng.choice(["nv21", "nv12", "yuv420p", "nv16"])
    sw, sh, dw, dh = rng.randint(200, 400), rng.randint(20, 64), rng.randint(200, 400), rng.randint(20, 64)
    fl = rng.choice([0x200, 0x4, 0x2]) | 0x80000
    return (sw, sh, sf, dw, dh, df, fl, rng.randint(0, 1 << 20), {"strip_min_w": 0}, [rng.randint(1, 5) for _ in range(3)])


def run(case, it):
    sw, sh, sf, dw, dh, df, flags, k, tune, sizes = case
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    p = SwsContext(sw, sh, sf, dw, dh, df, flags)
    for kk, v in list(tune.items()) + list(tune_env.items()):
        p.set_option(kk, v)
    seed = 1000 * k + 17 * it
    bad = 0
    for rnd, n in enumerate(sizes):
        refs, srcs, dsts = [], [], []
        for i in range(n):
            seed += 1
            s = OL.fill_random(OL.Frame(sf, sw, sh), seed)
            ref = OL.Frame(df, dw, dh, fill=0x33)
            assert o.scale(s, ref) >= 0
            refs.append(ref)
            hs = HostFrame(sf, sw, sh)
            for a, b in zip(hs.planes, s.planes):
                a[:] = b
            if rnd == 2 and (i & 1):
                hd = HostFrame(df, dw, dh)
                for a in hd.planes:
                    a[:] = 0x33
                srcs.append(hs); dsts.append(hd)
            else:
                dd = DeviceFrame(df, dw, dh); dd.buf.fill_(0x33)
                srcs.append(DeviceFrame(sf, sw, sh).upload(hs)); dsts.append(dd)
        torch.cuda.synchronize()
        assert p.scale_frames(srcs, dsts) == n
        p.sync()
        outs = [d.download() if isinstance(d, DeviceFrame) else d for d in dsts]
        for i in range(n):
            for pl, (a, b) in enumerate(zip(outs[i].planes, refs[i].planes)):
                rb = outs[i].row_bytes[pl]
                a, b = a[:, :rb], b[:, :rb]
                if np.array_equal(a, b):
                    continue
                bad += 1
                d = a != b
                rows = np.nonzero(d.any(axis=1))[0]; cols = np.nonzero(d.any(axis=0))[0]
                like = [j for j in range(n) if j != i and np.array_equal(a, refs[j].planes[pl][:, :rb])]
                # a second read of the same destination: did the bytes change after the first download (a late writer)?
                again = dsts[i].download().planes[pl][:, :rb] if isinstance(dsts[i], DeviceFrame) else a
                print(f"MISMATCH tag={tag} it={it} case={case[:7]} path={p.path()} call={rnd} n={n} frame={i} host={not isinstance(dsts[i], DeviceFrame)} plane={pl} "
                      f"bytes={int(d.sum())}/{d.size} rows={rows.min()}..{rows.max()} ({len(rows)}) cols={cols.min()}..{cols.max()} ({len(cols)}) "
                      f"still_fill={int((a[d] == 0x33).sum())} equals_other_frame={like} second_read_equal_ref={bool(np.array_equal(again, b))} "
                      f"got={a[d][:12].tolist()} want={b[d][:12].tolist()}", flush=True)
    p.close()
    return bad


t0 = time.time(); it = 0; nbad = 0
while time.time() - t0 < secs:
    nbad += run(draw(), it); it += 1
print(f"done tag={tag} mode={mode} iterations={it} mismatching_planes={nbad} tune={tune_env}", flush=True)
