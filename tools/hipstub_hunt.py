#!/usr/bin/env python3
"""The random conversions of tests/test_gpu_random.py through the product's whole HOST side on a box without a GPU: context, planner, device states, table blocks and
uploads, frame-table ring, staging of host frames, slices, batches sharded over several (fake) GPUs, teardown -- over tests/hipstub (a test double of the HIP runtime:
"device" memory is bounds-checked host memory, kernel launches are validated, logged and dropped, NOTHING is computed).  Run it with the stub preloaded, and under the
sanitizer builds to hunt the wild writer of DESIGN.md 8 in the code ASan could not reach without a GPU:

    LD_PRELOAD=tests/hipstub/libhipstub.so HIPSTUB_DEVICES=4 python tools/hipstub_hunt.py <N per generator> <seed>
    ... HIPSTUB_DEFER=1 ...: the laziest GPU the API allows -- queued work runs only when the host forces it (tests/hipstub/hipstub.cpp)
    tools/asan_env.sh env HIPSTUB_DEVICES=4 python tools/hipstub_hunt.py <N> <seed>          (asan_env.sh adds the stub when SWS_HIPSTUB=1)

What it checks: no call fails that the context accepted, every launch is well-formed, no copy or memset leaves its device allocation (the stub aborts), the table blocks
read back equal to what was uploaded after the conversions (sws_hip_debug_check), batches launch on the GPU that owns their frames, and every device allocation is
returned when the contexts are closed.  Pixels are not checked: there are none."""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SWS_RANDOM_N", "1")
import test_gpu_random as R  # noqa: E402
from librempeg_amd import swscale as S  # noqa: E402

STUB = C.CDLL(None)
if not hasattr(STUB, "hipstub_launches"):
    sys.exit("tools/hipstub_hunt.py: tests/hipstub/libhipstub.so is not preloaded (see the usage in this file's header)")
for f in ("hipstub_launches", "hipstub_copies", "hipstub_live_blocks", "hipstub_checked_pointers", "hipstub_unchecked_args", "hipstub_deferred_ops", "hipstub_pinned_checked", "hipstub_pending_ops"):
    getattr(STUB, f).restype = C.c_ulong
STUB.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
STUB.hipFree.argtypes = [C.c_void_p]
NDEV = C.c_int()
STUB.hipGetDeviceCount(C.byref(NDEV))
NDEV = NDEV.value


class StubFrame:
    """a picture in one "device" block of the stub, on a given ordinal: rows `pad` bytes longer than an aligned row, planes starting `shift` bytes off a 256-byte boundary,
    optionally bottom-up (negative linesize)"""

    def __init__(self, fmt, w, h, device=0, pad=0, shift=0, flip=False, fill=0x5A):
        self.fmt, self.w, self.h, self.device = fmt, w, h, device
        lay = S.plane_layout(fmt, w, h)
        self.nplanes = len(lay)
        self.linesize, self.offset, total = [], [], 0
        for rb, rows in lay:
            ls = ((rb + 63) // 64 * 64 if not pad else rb) + pad
            total = (total + 255) // 256 * 256 + shift
            self.offset.append(total)
            self.linesize.append(ls)
            total += ls * rows
        self.rows = [r for _, r in lay]
        self.flip = flip
        assert STUB.hipSetDevice(device) == 0
        p = C.c_void_p()
        assert STUB.hipMalloc(C.byref(p), total + 64) == 0
        self.base = p.value
        C.memset(self.base, fill, total + 64)

    def ptrs(self):
        p, s = (C.c_void_p * 4)(), (C.c_int * 4)()
        for i in range(self.nplanes):
            if self.flip:
                p[i] = self.base + self.offset[i] + (self.rows[i] - 1) * self.linesize[i]
                s[i] = -self.linesize[i]
            else:
                p[i] = self.base + self.offset[i]
                s[i] = self.linesize[i]
        return p, s

    def view(self):
        v = S.SwsFrameView()
        p, s = self.ptrs()
        for i in range(self.nplanes):
            v.data[i] = p[i]
            v.linesize[i] = s[i]
        v.width, v.height, v.format = self.w, self.h, S.PIX_FMT[self.fmt]
        return v

    def free(self):
        if self.base:
            STUB.hipFree(self.base)
            self.base = None


def host_frame(fmt, w, h):
    f = S.HostFrame(fmt, w, h)
    for a in f.planes:
        a[:] = 0x5A
    f.free = lambda: None
    return f


def slice_ptrs(frame, fmt, y0):
    lay = S._FORMATS[fmt] if hasattr(S, "_FORMATS") else None
    p, s = frame.ptrs()
    q = (C.c_void_p * 4)()
    kind, lh = (lay[1], lay[3]) if lay else ("planar", 1)
    for i in range(frame.nplanes):
        rows = y0 if (i == 0 or i == 3 or kind in ("rgbp", "packed", "gray")) else (y0 >> lh)
        if kind == "pal" and i == 1:
            rows = 0
        q[i] = p[i] + rows * s[i]
    return q, s


def plan_digest(p):
    dg = (C.c_uint64 * 3)()
    p.L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    return (p.L.sws_hip_plan(p.c, dg), dg[0])


def parts(c):
    opts = next((x for x in c[7:] if isinstance(x, dict) and ("dither" in x or "src_range" in x or "threads" in x)), None)
    tune = next((x for x in c[7:] if isinstance(x, dict) and x is not opts and any(k.startswith(("strip_", "no_")) for k in x)), None)
    cs = next((x for x in c[7:] if isinstance(x, tuple) and len(x) == 7 and all(isinstance(v, int) for v in x)), None)
    return opts, tune, cs


TUNE = ["strip_min_w", "strip_cols_l", "strip_cols_c", "no_mixed", "no_wave", "no_march", "no_strip", "no_strip_dma", "no_dot2", "no_tile", "no_strip_short", "no_strip_dma8",
        "no_strip_rgbsrc", "no_rgbsrc2", "no_fast_banks", "no_short_forms", "no_strip_wide", "no_generic_kinds", "no_layout_stream"]
TUNE_VALUES = {"strip_min_w": [0, 64, 320], "strip_cols_l": [2, 4], "strip_cols_c": [1, 2]}


def interleaved(n, seed, rng, stats, failures):
    """several contexts alive at once, used in turn with fresh and with kept frames on any GPU, re-planned under other launch options between conversions, checked and
    closed in any order: what a process that holds many scalers does (and what pytest -n 4 does to one GPU).  A table pointer kept across a regrown / freed block, a
    frame table that outlives its frames' ring span, a peer state freed twice shows as a stub abort"""
    cases = R._strip_cases(n, seed) + R._cases(n, seed + 1) + R._batch_cases(n, seed + 2) + R._strip_cases(n, seed + 3, R.R4_SRC, R.R4_DST)
    pool = []
    done = 0
    for step in range(6 * n):
        op = rng.random()
        if not pool or (op < 0.3 and len(pool) < 8):
            c = rng.choice(cases)
            opts, tune, cs = parts(c)
            try:
                p = S.SwsContext(*c[:7], **(opts or {}))
            except Exception:
                continue
            for k, v in (tune or {}).items():
                p.set_option(k, v)
            if cs and p.set_colorspace(*cs) < 0:
                p.close()
                continue
            pool.append(dict(p=p, c=c, frames=[], pairs=[], tune=dict(tune or {}), opts=opts, cs=cs))
        elif op < 0.82:
            e = rng.choice(pool)
            p, c = e["p"], e["c"]
            sw, sh, sf, dw, dh, df = c[:6]
            if rng.random() < 0.15:
                k = rng.choice(TUNE)
                e["tune"][k] = rng.choice(TUNE_VALUES.get(k, [0, 1]))
                p.set_option(k, e["tune"][k])                                            # the next conversion re-plans
            nb = rng.randint(1, 4)
            pairs = []
            for i in range(nb):
                if e["pairs"] and rng.random() < 0.5:
                    pairs.append(rng.choice(e["pairs"]))                                 # frames this context has seen: cached frame tables
                else:
                    g = rng.choice([-1] + list(range(NDEV)))
                    s = host_frame(sf, sw, sh) if g < 0 else StubFrame(sf, sw, sh, g)
                    d = host_frame(df, dw, dh) if g < 0 else StubFrame(df, dw, dh, g)
                    e["frames"] += [s, d]
                    e["pairs"].append((s, d))
                    pairs.append((s, d))
            if len({id(d) for _, d in pairs}) < len(pairs):
                pairs = pairs[:1]
            r = p.scale_frames([s for s, _ in pairs], [d for _, d in pairs]) if (len(pairs) > 1 or rng.random() < 0.5) else p.scale(*pairs[0])
            done += 1
            stats["total"] += 1
            if r > 0 and rng.random() < 0.25:
                # what this long-lived, re-planned context now holds against a FRESH context given the same options and the same frames: the path, the kernel
                # and the contents of every table block (digest 0 of sws_hip_plan: no addresses in it)
                q = S.SwsContext(*c[:7], **(e["opts"] or {}))
                for k, v in e["tune"].items():
                    q.set_option(k, v)
                if e["cs"]:
                    q.set_colorspace(*e["cs"])
                if len(pairs) > 1:
                    q.scale_frames([s for s, _ in pairs], [d for _, d in pairs])
                else:
                    q.scale(*pairs[0])
                a, b = (p.path(), p.kernel_name(), plan_digest(p)), (q.path(), q.kernel_name(), plan_digest(q))
                stats["compared"] = stats.get("compared", 0) + 1
                if a != b:
                    stats["stale_state"] = stats.get("stale_state", 0) + 1
                    failures.append(("a long-lived context differs from a fresh one", c[:7], e["tune"], a, b))
                q.close()
            if r != (len(pairs) if (len(pairs) > 1 or r == 1) else dh) and r != dh:
                stats["failed_calls"] += 1
                failures.append(("call failed (interleaved)", c[:7], r, p.path()))
        elif op < 0.9:
            e = rng.choice(pool)
            bad, text = e["p"].debug_check()
            if bad:
                stats["bad_tables"] += 1
                failures.append(("table blocks differ from their uploads (interleaved)", e["c"][:7], text))
        else:
            e = pool.pop(rng.randrange(len(pool)))
            e["p"].sync()
            e["p"].close()
            for f in e["frames"]:
                f.free()
    for e in pool:
        e["p"].close()
        for f in e["frames"]:
            f.free()
    print(f"interleaved contexts: {done} conversions, {STUB.hipstub_launches()} launches and {STUB.hipstub_copies()} copies so far", flush=True)


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed ^ 0xB0B)
    stats = dict(total=0, refused=0, failed_calls=0, bad_tables=0, wrong_gpu=0)
    failures = []
    gens = [("conversions", R._cases(n, seed)), ("options", R._opt_cases(n, seed + 1)), ("strip family", R._strip_cases(n, seed + 2)),
            ("round-4 routes", R._strip_cases(n, seed + 3, R.R4_SRC, R.R4_DST)), ("few rows", R._short_cases(n, seed + 4)), ("batches", R._batch_cases(n, seed + 5)),
            ("round-4 batches", R._batch_cases(n, seed + 8, R.R4_SRC, R.R4_DST)), ("unaligned", R._odd_cases(n, seed + 6)), ("slice sequences", R._slice_cases(n, seed + 7))]
    logpath = os.environ.get("HIPSTUB_LOG")
    for name, cases in gens:
        for c in cases:
            sw, sh, sf, dw, dh, df, flags = c[:7]
            opts, tune, cs = parts(c)
            stats["total"] += 1
            try:
                p = S.SwsContext(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
            except Exception:
                stats["refused"] += 1
                continue
            frames = []
            try:
                for k, v in (tune or {}).items():
                    p.set_option(k, v)
                if cs and p.set_colorspace(*cs) < 0:
                    stats["refused"] += 1
                    continue
                pad = shift = 0
                flip = 0
                if name == "unaligned":
                    pad, shift, flip = c[-1]
                    if any(x in sf + df for x in ("16", "10", "12", "14", "9", "48", "64", "f32", "xyz", "p0", "p2", "p4", "y2", "xv", "x2")):
                        pad, shift = pad & ~3, shift & ~3

                def mk(fmt, w, h, g, is_src):
                    if g < 0:
                        f = host_frame(fmt, w, h)
                    else:
                        f = StubFrame(fmt, w, h, g, pad, shift, bool(flip & (1 if is_src else 2)))
                    frames.append(f)
                    return f
                rets, want = [], []
                if name in ("batches", "round-4 batches"):
                    for rnd, nb in enumerate(c[9]):
                        srcs, dsts, owners = [], [], []
                        for i in range(nb):
                            g = -1 if (rnd == 2 and (i & 1)) else rng.randrange(NDEV)
                            srcs.append(mk(sf, sw, sh, g, True)); dsts.append(mk(df, dw, dh, g, False)); owners.append(g)
                        if logpath:
                            mark = os.path.getsize(logpath) if os.path.exists(logpath) else 0
                        rets.append(p.scale_frames(srcs, dsts)); want.append(nb)
                        if logpath and rets[-1] == nb:
                            with open(logpath) as fh:
                                fh.seek(mark)
                                used = {int(ln.split("dev=")[1].split()[0]) for ln in fh if ln.startswith("launch ")}
                            # (frames in host memory are dealt round-robin over the GPUs, csrc/dev_exec.hip sws_hip_plan_shards: only all-HBM batches pin the set)
                            if -1 not in owners and not used <= set(owners):
                                stats["wrong_gpu"] += 1
                                failures.append(("launches on a GPU that owns no frame of the batch", c[:7], sorted(used), owners))
                elif name == "slice sequences":
                    nsl = c[9]
                    srng = random.Random(c[7])
                    cuts = sorted({4 * srng.randint(1, max(1, sh // 4 - 1)) for _ in range(nsl - 1)} | {0, sh})
                    cuts = [x for x in cuts if x <= sh]
                    g = rng.randrange(NDEV)
                    s, d = mk(sf, sw, sh, g, True), mk(df, dw, dh, g, False)
                    dp, dstr = d.ptrs()
                    for y0, y1 in zip(cuts[:-1], cuts[1:]):
                        sp, ss = slice_ptrs(s, sf, y0)
                        rets.append(p.L.sws_scale(p.c, sp, ss, y0, y1 - y0, dp, dstr)); want.append(None)
                else:
                    g = rng.choice([-1] + list(range(NDEV)))
                    s, d = mk(sf, sw, sh, g, True), mk(df, dw, dh, g, False)
                    for _ in range(1 + (stats["total"] % 3 == 0)):                       # a sample converts twice on the same context: cached tables, ring reuse
                        rets.append(p.scale(s, d)); want.append(dh)
                p.sync()
                if any(r < 0 or (w is not None and r != w) for r, w in zip(rets, want)):
                    stats["failed_calls"] += 1
                    failures.append(("call failed", name, c[:7], rets, p.path()))
                bad, text = p.debug_check()
                if bad:
                    stats["bad_tables"] += 1
                    failures.append(("table blocks differ from their uploads", c[:7], text))
            finally:
                p.close()
                for f in frames:
                    f.free()
        print(f"{name}: {len(cases)} cases, {STUB.hipstub_launches()} launches and {STUB.hipstub_copies()} copies so far", flush=True)
    interleaved(n, seed + 9, rng, stats, failures)
    live = STUB.hipstub_live_blocks()
    for f in failures[:40]:
        print("FAIL", f, flush=True)
    print(f"kernel-argument pointers checked at launch: {STUB.hipstub_checked_pointers()} (arguments of plan-struct types the stub does not know: {STUB.hipstub_unchecked_args()}); "
          f"HIPSTUB_DEFER: {STUB.hipstub_deferred_ops()} operations ran late, {STUB.hipstub_pinned_checked()} copies from pinned memory compared with what was queued, "
          f"{STUB.hipstub_pending_ops()} still queued")
    print(f"{stats}, device blocks still allocated after every context was closed: {live} ")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
