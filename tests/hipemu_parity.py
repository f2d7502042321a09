#!/usr/bin/env python3
"""The x86 EMULATION build of the product (tests/hipemu: the library's own sources, kernels included, compiled for the host and run thread by thread over the HIP runtime
test double) against the oracle, on the box without a GPU.  Test infrastructure: it says that the C++ of the kernels and the host side around them compute the
oracle's pictures for the cases drawn -- after a round of host restructuring without a GPU, that is worth knowing -- and, built with AddressSanitizer, that no kernel
reads or writes outside the blocks it was given.  It is NOT the gfx950 code object and proves nothing about it: GPU parity is the -m gpu suite's business.

    make -C tests/hipstub && make -C tests/hipemu
    LD_PRELOAD=tests/hipstub/libhipstub.so SWS_HIP_LIBRARY=tests/hipemu/libswscale_hip_emu.so SWS_HIP_NO_TORCH=1 python tests/hipemu_parity.py <N per generator> <seed> [generator ...]
"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SWS_RANDOM_N", "1")
import numpy as np  # noqa: E402
import oracle_lib as OL  # noqa: E402
import test_gpu_random as R  # noqa: E402
_argv, sys.argv = sys.argv, ["x", "1", "1"]
import hipstub_hunt as H  # noqa: E402  (StubFrame, the stub's counters)
sys.argv = _argv
from librempeg_amd import swscale as S  # noqa: E402

FILL = 0x5A


def to_stub(frame, oracle_frame):
    """the oracle frame's visible bytes into the stub "device" frame"""
    for i, (a, rb) in enumerate(zip(oracle_frame.planes, oracle_frame.row_bytes)):
        for y in range(a.shape[0]):
            row = np.ascontiguousarray(a[y, :rb])
            yy = (frame.rows[i] - 1 - y) if frame.flip else y
            C.memmove(frame.base + frame.offset[i] + yy * frame.linesize[i], row.ctypes.data, rb)


def from_stub(frame, like):
    out = []
    for i, (a, rb) in enumerate(zip(like.planes, like.row_bytes)):
        buf = np.empty((a.shape[0], rb), dtype=np.uint8)
        for y in range(a.shape[0]):
            yy = (frame.rows[i] - 1 - y) if frame.flip else y
            C.memmove(buf[y].ctypes.data, frame.base + frame.offset[i] + yy * frame.linesize[i], rb)
        out.append(buf)
    return out


def run_case(c, name, rng, stats, failures, max_pixels):
    sw, sh, sf, dw, dh, df, flags = c[:7]
    if sw * sh > max_pixels or dw * dh > max_pixels:
        stats["too large"] += 1
        return
    opts, tune, cs = H.parts(c)
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
    except Exception:
        stats["oracle refuses"] += 1
        return
    try:
        p = S.SwsContext(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
    except Exception:
        stats["product refuses"] += 1
        return
    frames = []
    try:
        for k, v in (tune or {}).items():
            p.set_option(k, v)
        if cs:
            rc = o.set_colorspace(*cs)
            if rc != p.set_colorspace(*cs):
                failures.append(("set_colorspace answers differ", c[:7]))
                return
            if rc < 0:
                return
        nframes = 1 + (name == "batches") * rng.randint(0, 2)
        srcs, dsts, refs = [], [], []
        pad, shift, flip = (c[-1] if name == "unaligned" else (0, 0, 0))
        if name == "unaligned" and any(x in sf + df for x in ("16", "10", "12", "14", "9", "48", "64", "f32", "xyz", "p0", "p2", "p4", "y2", "xv", "x2")):
            pad, shift = pad & ~3, shift & ~3
        for n in range(nframes):
            src = OL.fill_random(OL.Frame(sf, sw, sh), c[7] * 7 + n + 11)
            ref = OL.Frame(df, dw, dh, fill=FILL)
            if o.scale(src, ref) < 0:
                stats["oracle refuses"] += 1
                return
            s = H.StubFrame(sf, sw, sh, 0, pad, shift, bool(flip & 1), fill=0)
            d = H.StubFrame(df, dw, dh, 0, pad, shift, bool(flip & 2), fill=FILL)
            frames += [s, d]
            to_stub(s, src)
            srcs.append(s); dsts.append(d); refs.append(ref)
        t0 = time.time()
        r = p.scale_frames(srcs, dsts) if nframes > 1 else p.scale(srcs[0], dsts[0])
        p.sync()
        stats["seconds in the product"] += time.time() - t0
        path, kern = p.path(), p.kernel_name()
        if r < 0:
            stats["product call fails"] += 1
            failures.append(("call failed", c[:7], r, path))
            return
        stats["compared"] += 1
        stats.setdefault("paths", {}).setdefault(path, [0, 0])[0] += 1
        for n in range(nframes):
            got = from_stub(dsts[n], refs[n])
            for i, (a, b, rb) in enumerate(zip(got, refs[n].planes, refs[n].row_bytes)):
                b = b[:, :rb]
                if df in ("monob", "monow") and (dw & 7):
                    a, b = a.copy(), b.copy()
                    m = (0xFF00 >> (dw & 7)) & 0xFF
                    a[:, rb - 1] &= m; b[:, rb - 1] &= m
                if not np.array_equal(a, b):
                    stats["different"] += 1
                    stats["paths"][path][1] += 1
                    bad = np.argwhere(a != b)
                    failures.append(("DIFFERENT", name, c[:7], opts, tune, cs, path, kern, f"frame {n} plane {i}: {len(bad)} bytes, first at row {bad[0][0]} byte {bad[0][1]}"))
                    return
    finally:
        p.close()
        for f in frames:
            f.free()


def interleaved(n, seed, rng, stats, failures, max_pixels):
    """up to 6 contexts alive, converted with in turn, their launch options changed between conversions (the next call re-plans), closed in any order: every picture of
    every long-lived, re-planned context against the oracle"""
    cases = [c for c in R._strip_cases(n, seed) + R._cases(n, seed + 1) + R._strip_cases(n, seed + 2, R.R4_SRC, R.R4_DST) if c[0] * c[1] <= max_pixels and c[3] * c[4] <= max_pixels]
    pool = []
    for step in range(5 * n):
        op = rng.random()
        if not pool or (op < 0.25 and len(pool) < 6):
            c = rng.choice(cases)
            opts, tune, cs = H.parts(c)
            try:
                o = OL.Oracle(*c[:7], **(opts or {}))
                p = S.SwsContext(*c[:7], **(opts or {}))
            except Exception:
                continue
            for k, v in (tune or {}).items():
                p.set_option(k, v)
            if cs and (o.set_colorspace(*cs) < 0 or p.set_colorspace(*cs) < 0):
                p.close()
                continue
            pool.append(dict(p=p, o=o, c=c, n=0))
        elif op < 0.9:
            e = rng.choice(pool)
            p, o, c = e["p"], e["o"], e["c"]
            sw, sh, sf, dw, dh, df = c[:6]
            if rng.random() < 0.3:
                k = rng.choice(H.TUNE)
                p.set_option(k, rng.choice(H.TUNE_VALUES.get(k, [0, 1])))
            e["n"] += 1
            src = OL.fill_random(OL.Frame(sf, sw, sh), c[7] * 13 + e["n"])
            ref = OL.Frame(df, dw, dh, fill=FILL)
            if o.scale(src, ref) < 0:
                continue
            s, d = H.StubFrame(sf, sw, sh, 0, fill=0), H.StubFrame(df, dw, dh, 0, fill=FILL)
            to_stub(s, src)
            r = p.scale(s, d)
            p.sync()
            stats["compared"] += 1
            path = p.path()
            stats.setdefault("paths", {}).setdefault(path, [0, 0])[0] += 1
            if r < 0:
                stats["product call fails"] += 1
                failures.append(("call failed (interleaved)", c[:7], r, path))
            else:
                got = from_stub(d, ref)
                for i, (a, b, rb) in enumerate(zip(got, ref.planes, ref.row_bytes)):
                    b = b[:, :rb]
                    if df in ("monob", "monow") and (dw & 7):
                        a, b = a.copy(), b.copy()
                        m = (0xFF00 >> (dw & 7)) & 0xFF
                        a[:, rb - 1] &= m; b[:, rb - 1] &= m
                    if not np.array_equal(a, b):
                        stats["different"] += 1
                        stats["paths"][path][1] += 1
                        failures.append(("DIFFERENT (a long-lived, re-planned context)", c[:7], path, p.kernel_name(), f"conversion {e['n']} plane {i}: {int(np.count_nonzero(a != b))} bytes"))
                        break
            s.free(); d.free()
        else:
            pool.pop(rng.randrange(len(pool)))["p"].close()
    for e in pool:
        e["p"].close()


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    want = sys.argv[3:]
    max_pixels = int(os.environ.get("HIPEMU_MAX_PIXELS", "200000"))
    rng = random.Random(seed)
    stats = {k: 0 for k in ("compared", "different", "oracle refuses", "product refuses", "product call fails", "too large", "seconds in the product")}
    failures = []
    gens = [("conversions", lambda: R._cases(n, seed)), ("options", lambda: R._opt_cases(n, seed + 1)), ("strip family", lambda: R._strip_cases(n, seed + 2)),
            ("round-4 routes", lambda: R._strip_cases(n, seed + 3, R.R4_SRC, R.R4_DST)), ("few rows", lambda: R._short_cases(n, seed + 4)),
            ("batches", lambda: R._batch_cases(n, seed + 5)), ("unaligned", lambda: R._odd_cases(n, seed + 6))]
    for name, make in gens:
        if want and name not in want:
            continue
        for c in make():
            run_case(c, name, rng, stats, failures, max_pixels)
        print(f"{name}: {stats['compared']} compared so far, {stats['different']} different", flush=True)
    if not want or "interleaved" in want:
        interleaved(n, seed + 9, rng, stats, failures, max_pixels)
        print(f"interleaved contexts: {stats['compared']} compared so far, {stats['different']} different", flush=True)
    for f in failures[:60]:
        print("FAIL", f, flush=True)
    paths = stats.pop("paths", {})
    for k in sorted(paths):
        print(f"   {k}: {paths[k][0]} compared, {paths[k][1]} different")
    E = S.load_library()
    extra = ""
    if hasattr(E, "hipemu_threads"):
        E.hipemu_threads.restype = E.hipemu_launches.restype = C.c_ulong
        extra = f"; {E.hipemu_launches()} launches, {E.hipemu_threads()} kernel threads executed"
        if hasattr(E, "hipemu_oob_report"):
            buf = C.create_string_buffer(1 << 16)
            if E.hipemu_oob_report(buf, len(buf)):
                print("buffer LOADS that left their device block (HIPEMU_CHECK_BUFFERS=1; kernel, dwords, most bytes past the end):")
                for ln in buf.value.decode().splitlines():
                    print("   ", ln)
    print(f"{stats}{extra}")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
