#!/usr/bin/env python3
"""bench.py -- sws_scale throughput of libswscale_hip on MI355X.

A "step" is ONE sws_scale_frames() call = one pass of the hot path over one batch of synthetic
HBM-resident frames (one kernel launch on the dominant path).  Default workload = BASELINE.json
configs[1]: 3840x2160 yuv420p -> rgb24, SWS_BICUBIC|SWS_BITEXACT.  With those flags and equal sizes
the reference selects the unscaled LUT converter yuv2rgb_c_24_rgb and so does this
library ("c2a").  The polyphase variant of the same conversion (flags + SWS_ACCURATE_RND, "c2b":
1-tap luma, 4-tap bicubic vertical chroma, yuv2rgb24_X) is measured too and reported in the same
JSON line under "variants".  Both have the same algorithmic bytes (37 324 800 B / frame).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run,
one rank per GPU (RCCL).  Frames are sharded across ranks (weak scaling, no data-path collective);
the only collective is the one-off broadcast of the context's table blob from rank 0.
Rank 0 prints ONE JSON line.

`python bench.py --gpus N --inproc` (no launcher) measures the OTHER multi-GPU form: ONE process, ONE context, ONE
sws_scale_frames() call per step over frames that live on N GPUs -- the library shards them itself (a frame is converted on the
GPU that holds it; dev_exec.hip: dev_run), each GPU gets its own copy of the tables at first use, launches go out on every GPU's
stream before anything is waited for.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from librempeg_amd import (SwsContext, DeviceFrame, HostFrame, plane_layout,  # noqa: E402
                           SWS_BICUBIC, SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_CS_BT2020)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags, colorspace, default batch, description)
    "c2a": (3840, 2160, "yuv420p", 3840, 2160, "rgb24", SWS_BICUBIC | SWS_BITEXACT, None, 32,
            "C2a 3840x2160 yuv420p->rgb24 SWS_BICUBIC|SWS_BITEXACT (reference path: unscaled yuv2rgb_c_24_rgb)"),
    "c2b": (3840, 2160, "yuv420p", 3840, 2160, "rgb24", SWS_BICUBIC | SWS_BITEXACT | SWS_ACCURATE_RND, None, 32,
            "C2b 3840x2160 yuv420p->rgb24 SWS_BICUBIC|SWS_BITEXACT|SWS_ACCURATE_RND (polyphase chain, 4-tap vertical chroma)"),
    "c1": (1280, 720, "yuv420p", 640, 360, "yuv420p", SWS_BILINEAR | SWS_BITEXACT, None, 256,
           "C1 1280x720->640x360 yuv420p SWS_BILINEAR|SWS_BITEXACT"),
    "c3a": (7680, 4320, "yuv420p10le", 7680, 4320, "p010le", SWS_LANCZOS | SWS_BITEXACT, None, 8,
            "C3a 7680x4320 yuv420p10le->p010le same size (reference path: planarToP01xWrapper)"),
    # (batches: a strip-kernel launch is banded for one resident round of 4096 waves, so the fewer frames a call holds the shorter its bands and
    #  the larger the share of the per-band ring fill -- C3b: 8 frames 0.44 - 0.47 of the HBM peak, 16 frames 0.51, 32 frames 0.52 - 0.56, 64 frames 0.59;
    #  D1: 64 frames 0.34 - 0.37, 128 frames 0.40; C1: 128 frames 0.24, 256 frames 0.25)
    "c3b": (7680, 4320, "yuv420p10le", 3840, 2160, "p010le", SWS_LANCZOS | SWS_BITEXACT, None, 64,
            "C3b 7680x4320->3840x2160 yuv420p10le->p010le SWS_LANCZOS (12-tap h and v)"),
    "c4": (1920, 1080, "nv12", 1920, 1080, "bgr0", SWS_BICUBIC | SWS_BITEXACT, None, 64,
           "C4 1920x1080 nv12->bgr0 SWS_BICUBIC|SWS_BITEXACT (main path, 4-tap vertical chroma)"),
    "c5": (3840, 2160, "gbrpf32le", 3840, 2160, "yuv444p16le", SWS_BICUBIC | SWS_BITEXACT,
           (SWS_CS_BT2020, 1, SWS_CS_BT2020, 1), 8,
           "C5 3840x2160 gbrpf32le->yuv444p16le BT.2020 full range"),
    # not BASELINE.json configs: the most common real use of sws_scale, kept as bench variants
    "d1": (3840, 2160, "yuv420p", 1920, 1080, "rgb24", SWS_BICUBIC | SWS_BITEXACT, None, 128,
           "D1 3840x2160->1920x1080 yuv420p->rgb24 SWS_BICUBIC|SWS_BITEXACT (downscale to packed RGB)"),
    "d2": (3840, 2160, "yuv420p", 1920, 1080, "bgra", SWS_BICUBIC | SWS_BITEXACT, None, 128,
           "D2 3840x2160->1920x1080 yuv420p->bgra SWS_BICUBIC|SWS_BITEXACT (downscale to packed RGB)"),
    # the capture / render -> encoder shape (scaled packed RGB into 4:2:0): sws_k_strip_rgbsrc
    "e1": (1920, 1080, "rgb24", 1280, 720, "yuv420p", SWS_BICUBIC | SWS_BITEXACT, None, 64,
           "E1 1920x1080->1280x720 rgb24->yuv420p SWS_BICUBIC|SWS_BITEXACT (packed RGB source, downscale)"),
    "e2": (3840, 2160, "bgra", 1920, 1080, "yuv420p", SWS_BICUBIC | SWS_BITEXACT, None, 32,
           "E2 3840x2160->1920x1080 bgra->yuv420p SWS_BICUBIC|SWS_BITEXACT (packed RGB source, downscale)"),
    # round 5: MPEG <-> JPEG range conversion inside the strip kernels (an MJPEG camera's frames into an encoder's format; capture into a JPEG encoder's)
    "r1": (1920, 1080, "yuvj422p", 1920, 1080, "yuv420p", SWS_BICUBIC | SWS_BITEXACT, None, 64,
           "R1 1920x1080 yuvj422p->yuv420p SWS_BICUBIC|SWS_BITEXACT (full -> limited range, vertical chroma step)"),
    "r2": (1920, 1080, "bgra", 1920, 1080, "yuvj420p", SWS_BICUBIC | SWS_BITEXACT, None, 64,
           "R2 1920x1080 bgra->yuvj420p SWS_BICUBIC|SWS_BITEXACT (same size, into full range)"),
    # round 5: 19-bit intermediates -- decoder output into planar float RGB (inference input)
    "w1": (3840, 2160, "nv12", 1920, 1080, "gbrpf32le", SWS_BICUBIC | SWS_BITEXACT, None, 32,
           "W1 3840x2160->1920x1080 nv12->gbrpf32le SWS_BICUBIC|SWS_BITEXACT (decoder output -> planar float RGB)"),
    # round 5: the flags players and capture tools pass -- SWS_FAST_BILINEAR (ff_hyscale_fast_c as two-tap banks under the strip kernels) and bilinear up-scaling into
    # packed RGB (the writers' short vertical forms in the strip plan)
    "f1": (3840, 2160, "yuv420p", 1920, 1080, "yuv420p", SWS_FAST_BILINEAR, None, 64,
           "F1 3840x2160->1920x1080 yuv420p SWS_FAST_BILINEAR (the fast horizontal functions as two-tap banks)"),
    "u1": (1280, 720, "yuv420p", 1920, 1080, "bgra", SWS_BILINEAR, None, 64,
           "U1 1280x720->1920x1080 yuv420p->bgra SWS_BILINEAR (a player's up-scaling: yuv2rgb_2 rows in the strip plan)"),
}


# The strip-kernel variants are reported at TWO batch sizes: the workload's default above (large: the ring fill of a band and the launch
# tail are amortised) and the batch the round-1 / round-2 records quote, so that numbers stay comparable from round to round.
SMALL_BATCH = {"c3b": 8, "d1": 32, "d2": 32, "c1": 64}
# what the default run (`--variants auto` next to the headline c2a) times: EVERY BASELINE.json configuration first (c2b = the polyphase twin of
# configs[1], c3a / c3b = configs[2] same size / scaled, c4 = configs[3] at one GPU's 64-frame share of the 512-frame batch, c5 = configs[4],
# c1 = configs[0]), then the strip-family workloads the rounds' reviews follow
AUTO_VARIANTS = ["c2b", "c3a", "c3b", "c4", "c5", "c1", "d1", "e2", "r1", "w1", "f1", "u1"]


def clamp_batch(name, batch, device_index):
    """frames per step that fit the GPU: source + destination pictures (and the largest per-frame working picture a helper pass may keep) must
    stay below 40 % of the free memory, so that the default bench run cannot exhaust a smaller or busier device."""
    sw, sh, sf, dw, dh, df = WORKLOADS[name][:6]
    per = algorithmic_bytes(sw, sh, sf, dw, dh, df) * 2 + 16 * dw * dh + (1 << 20)
    try:
        free, _ = torch.cuda.mem_get_info(device_index)
    except Exception:
        return batch
    return max(1, min(batch, int(0.4 * free / per)))


# what pins each workload's arithmetic to the reference (oracle/README.md)
PARITY_PIN = {
    "c2a": "LUT tables pinned by the reference's pixfmt MD5s (yuv2rgb24_X); the unscaled converter's 8/4/2-pixel block structure "
           "(yuv2rgb_c_24_rgb) by the oracle twin only: no reference golden exists without accurate_rnd",
    "c2b": "fate-pixfmt-rgb24 MD5 (tests/ref/pixfmt/rgb24: bicubic+accurate_rnd+bitexact) through oracle and HIP path",
    "c3a": "fate-pixfmt p010le / yuv420p10le MD5s", "c3b": "SURVEY Appendix D filter banks + fate-pixfmt p010le MD5s",
    "c4": "fate-pixfmt nv12 / bgr0 MD5s", "c5": "fate-sws-floatimg-cmp (tests/ref/fate/sws-floatimg-cmp) + SURVEY Appendix D rgb2yuv BT.2020",
    "c1": "fate-sws-yuv-range / filter-scalechroma framecrcs (hScale8To15_c, yuv2planeX_8_c)",
}


def algorithmic_bytes(sw, sh, sfmt, dw, dh, dfmt):
    """visible bytes of all planes in + out (no padding, tables or halo re-reads)."""
    return sum(rb * r for rb, r in plane_layout(sfmt, sw, sh)) + sum(rb * r for rb, r in plane_layout(dfmt, dw, dh))


def fill_device_frame(fr, seed):
    """synthetic content: uniform random bytes (10-bit / float formats get format-valid samples)."""
    g = torch.Generator(device=fr.buf.device)
    g.manual_seed(seed)
    f = fr.fmt
    for i in range(fr.nplanes):
        t = fr.plane_tensor(i)
        rows, ls = t.shape
        if f in ("yuv420p10le", "yuv444p10le"):
            v = torch.randint(0, 1024, (rows, ls // 2), generator=g, device=t.device, dtype=torch.int16)
            t.copy_(v.view(torch.uint8).view(rows, ls))
        elif f == "p010le":
            v = (torch.randint(0, 1024, (rows, ls // 2), generator=g, device=t.device, dtype=torch.int32) << 6).to(torch.int16)
            t.copy_(v.view(torch.uint8).view(rows, ls))
        elif f == "gbrpf32le":
            v = torch.rand((rows, ls // 4), generator=g, device=t.device, dtype=torch.float32) * 1.5 - 0.25
            t.copy_(v.view(torch.uint8).view(rows, ls))
        else:
            t.copy_(torch.randint(0, 256, (rows, ls), generator=g, device=t.device, dtype=torch.uint8))


TUNE = {}   # --opt name=value: launch heuristics passed to sws_hip_set_option() (A/B measurements; results never change)


def make_context(name, rank, world, device_index):
    ctx = _make_context(name, rank, world, device_index)
    for k, v in TUNE.items():
        ctx.set_option(k, v)
    return ctx


def _make_context(name, rank, world, device_index):
    sw, sh, sf, dw, dh, df, flags, cs, _, _ = WORKLOADS[name]
    if world == 1 or rank == 0:
        ctx = SwsContext(sw, sh, sf, dw, dh, df, flags, device=device_index)
        if cs:
            assert ctx.set_colorspace(*cs) >= 0
    else:
        ctx = SwsContext(sw, sh, sf, dw, dh, df, flags, device=device_index, empty=True)
    if world > 1:
        # one-off broadcast of the table blob (filters, LUT constants, plan) from rank 0 over RCCL/xGMI
        import torch.distributed as dist
        try:
            blob = ctx.export_tables() if rank == 0 else b""
            tdev = "cpu" if dist.get_backend() == "gloo" else f"cuda:{device_index}"
            n = torch.tensor([len(blob)], dtype=torch.int64, device=tdev)
            dist.broadcast(n, 0)
            buf = torch.empty(int(n.item()), dtype=torch.uint8, device=tdev)
            if rank == 0:
                buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
            dist.broadcast(buf, 0)
            if rank != 0:
                ctx.import_tables(bytes(buf.cpu().numpy().tobytes()))
        except Exception as e:  # the tables are a pure function of the options: every rank can also build them itself
            print(f"[bench rank {rank}] table broadcast failed ({e!r}); building the tables locally", file=sys.stderr, flush=True)
            if rank != 0:
                ctx.close()
                ctx = SwsContext(sw, sh, sf, dw, dh, df, flags, device=device_index)
                if cs:
                    assert ctx.set_colorspace(*cs) >= 0
    return ctx


def run_workload_inproc(name, batch, steps, warmup, ngpus):
    """one process, one context, frames on `ngpus` GPUs, one sws_scale_frames() call per step (in-library sharding)"""
    sw, sh, sf, dw, dh, df, flags, cs, dbatch, desc = WORKLOADS[name]
    batch = batch or dbatch
    ctx = make_context(name, 0, 1, 0)
    srcs, dsts = [], []
    for g in range(ngpus):
        for i in range(batch):
            s, d = DeviceFrame(sf, sw, sh, f"cuda:{g}"), DeviceFrame(df, dw, dh, f"cuda:{g}")
            fill_device_frame(s, 1000 * (g + 1) + i)
            srcs.append(s); dsts.append(d)
    for g in range(ngpus):
        torch.cuda.synchronize(g)
    b = ctx.make_batch(srcs, dsts)
    path = ctx.path()
    for _ in range(warmup):
        r = ctx.run_batch(b)
        assert r == batch * ngpus, f"sws_scale_frames returned {r}"
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_batch(b)
    ctx.sync()                       # waits for the context's streams on every GPU
    t1 = time.perf_counter()
    ctx.set_timing(True)             # kernel time of the home GPU's launch set (HIP events on its stream)
    kms = []
    for _ in range(min(steps, 10)):
        ctx.run_batch(b); ctx.sync()
        kms.append(ctx.last_kernel_ms())
    res = dict(name=name, desc=desc, batch=batch, wall_s=t1 - t0, kernel_ms_avg=float(np.mean(kms)), kernel_ms_min=float(np.min(kms)),
               path=path, kernel=ctx.kernel_name(), out_pixels_per_step=batch * ngpus * dw * dh,
               alg_bytes_per_step=batch * algorithmic_bytes(sw, sh, sf, dw, dh, df))
    del srcs, dsts
    ctx.close()
    torch.cuda.empty_cache()
    return res


def run_workload(name, batch, steps, warmup, rank, world, device_index, barrier):
    sw, sh, sf, dw, dh, df, flags, cs, dbatch, desc = WORKLOADS[name]
    batch = clamp_batch(name, batch or dbatch, device_index)
    dev = f"cuda:{device_index}"
    ctx = make_context(name, rank, world, device_index)
    stream = torch.cuda.Stream(device_index)  # a real (non-null) HIP stream shared by torch events and the library
    ctx.set_stream(stream.cuda_stream)
    srcs = [DeviceFrame(sf, sw, sh, dev) for _ in range(batch)]
    dsts = [DeviceFrame(df, dw, dh, dev) for _ in range(batch)]
    for i, s in enumerate(srcs):
        fill_device_frame(s, 1000 * (rank + 1) + i)
    torch.cuda.synchronize(device_index)
    b = ctx.make_batch(srcs, dsts)
    path = ctx.path()
    for _ in range(warmup):
        r = ctx.run_batch(b)
        assert r == batch, f"sws_scale_frames returned {r}"
    torch.cuda.synchronize(device_index)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    torch.cuda.synchronize(device_index)
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record(stream)
        ctx.run_batch(b)
        ev[k][1].record(stream)
    torch.cuda.synchronize(device_index)
    barrier()
    t1 = time.perf_counter()
    kernel_ms = [a.elapsed_time(c) for a, c in ev]
    res = dict(name=name, desc=desc, batch=batch, wall_s=t1 - t0, kernel_ms_avg=float(np.mean(kernel_ms)),
               kernel_ms_min=float(np.min(kernel_ms)), kernel_ms_median=float(np.median(kernel_ms)), path=path, kernel=ctx.kernel_name(),
               out_pixels_per_step=batch * dw * dh, alg_bytes_per_step=batch * algorithmic_bytes(sw, sh, sf, dw, dh, df))
    del srcs, dsts
    ctx.close()
    torch.cuda.empty_cache()
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# port / reference, one thread: profiles/ref_vs_port.json holds, per BASELINE configuration, the ms per frame of a C-only build of the REAL
# reference (configure --disable-asm, built under /tmp in the build container, never shipped) and of oracle/ ("the port") in the SAME
# container on one thread, best of N, as tools/ref_vs_port.sh measured them.  The GPU box has no reference, so the line this bench prints
# times the port there and carries (a) that ratio and (b) the port's rate scaled by it as the estimate of the reference's C path on this host.
def ref_vs_port():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "ref_vs_port.json")))
    except Exception:
        return {}


def cpu_leg(name, seconds, nthreads_list):
    """The oracle ("port" of the reference C path) timed on this host for one workload: frame-parallel with one context per thread
    (sws_scale itself never multi-threads).  Returns {threads: (frames, Mpix/s, ms per frame and thread)}."""
    import oracle_lib as OL
    sw, sh, sf, dw, dh, df, flags, cs, _, desc = WORKLOADS[name]
    OL.lib()

    def worker(out, idx, deadline):
        try:    # one thread per host core, pinned: the frame-parallel form the reference would be run in (one context per core)
            os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[idx % len(os.sched_getaffinity(0))]})
        except Exception:
            pass
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
        if cs:
            o.set_colorspace(*cs)
        src = OL.fill_random(OL.Frame(sf, sw, sh), 77 + idx)
        dst = OL.Frame(df, dw, dh)
        o.scale(src, dst)  # warm-up
        n = 0
        t0 = time.perf_counter()
        while True:
            o.scale(src, dst)
            n += 1
            if time.perf_counter() >= deadline and n >= 2:
                break
        out[idx] = (n, time.perf_counter() - t0)

    res = {}
    for nthreads in nthreads_list:
        out = [None] * nthreads
        deadline = time.perf_counter() + seconds
        th = [threading.Thread(target=worker, args=(out, i, deadline)) for i in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        frames = sum(n for n, _ in out)
        rate = sum(n / dt for n, dt in out)          # per-thread rates summed (each thread times its own loop; setup / fill excluded)
        res[nthreads] = (frames, rate * dw * dh / 1e6, 1e3 * nthreads / rate)
    return res


def cpu_baseline(name, seconds=6.0, others=("c1", "c2b", "c3a", "c3b", "c4", "c5"), other_seconds=1.5):
    """cpu_baseline of the JSON line: the headline workload on 1 thread and on all host cores, plus one short leg per other BASELINE
    configuration (C1 is BASELINE's CPU-only configuration)."""
    cores = os.cpu_count() or 1
    tl = [1, cores] if cores > 1 else [1]
    desc = WORKLOADS[name][9]
    rvp = ref_vs_port()

    def anchors(n, mp_all, mp_1):
        e = rvp.get(n)
        if not e:
            return {}
        k = e["port_over_reference"]
        return {"port_over_reference_1thread": k, "reference_c_ms_per_frame_1thread_build_container": e["reference_ms"],
                "port_ms_per_frame_1thread_build_container": e["port_ms"],
                "reference_c_estimate_all_threads": round(mp_all * k, 2), "reference_c_estimate_1thread": round(mp_1 * k, 2)}

    r = cpu_leg(name, seconds, tl)
    f1, mp1, ms1 = r[1]
    fN, mpN, msN = r[tl[-1]]
    configs = {}
    for o in others:
        if o == name:
            continue
        # (the 8K workloads hold 125 - 200 MB of pictures per thread: at most 32 threads, so that a many-core host is not asked for tens of GB)
        tlo = [1, min(tl[-1], 32)] if o in ("c3a", "c3b") and tl[-1] > 1 else tl
        ro = cpu_leg(o, other_seconds, tlo)
        configs[o] = {"workload": WORKLOADS[o][9], "value_1thread": round(ro[1][1], 2), "ms_per_frame_1thread": round(ro[1][2], 2),
                      "value_all_threads": round(ro[tlo[-1]][1], 2), "threads": tlo[-1], "frames_timed": [ro[1][0], ro[tlo[-1]][0]], "unit": "Mpixels/s"}
        configs[o].update(anchors(o, ro[tlo[-1]][1], ro[1][1]))
    out = {"value": round(mpN, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
           "value_1thread": round(mp1, 2), "ms_per_frame_1thread": round(ms1, 2)}
    out.update(anchors(name, mpN, mp1))
    out["ratio_source"] = rvp.get("_source")
    out["configs"] = configs
    out["sample"] = (f"{desc}; oracle/ = scalar C restatement of the reference's C path (gcc -O3 -fno-tree-vectorize like the reference's "
                     f"own flags), ~{seconds:.0f}s per leg: {f1} frames on 1 thread ({ms1:.2f} ms/frame), {fN} frames on {cores} threads "
                     f"(one context per thread, pinned).  `value` is the PORT on this host.  `port_over_reference_1thread` = port ms / real "
                     f"reference ms (C-only build, no SIMD) measured side by side in the build container; `reference_c_estimate_*` = the port's "
                     f"rate here x that ratio.  Neither is the reference's hand-written x86 SIMD, which would be faster again.  `configs`: the "
                     f"other BASELINE configurations, {other_seconds:.1f}s per leg")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2a", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = workload default)")
    ap.add_argument("--variants", default="auto", help="comma list of extra workloads to time (auto: c2b when workload is c2a)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--opt", action="append", default=[], help="name=value launch heuristic (sws_hip_set_option), repeatable")
    ap.add_argument("--rccl", action="store_true", help="with --inproc: the peers' tables by ncclBroadcast from the home GPU (option rccl_tables) instead of one H2D copy per GPU")
    ap.add_argument("--inproc", action="store_true", help="N GPUs from ONE process: in-library sharding of sws_scale_frames() (no launcher)")
    args = ap.parse_args()
    for o in args.opt:
        k, v = o.split("=")
        TUNE[k] = int(v)
    if args.rccl:
        TUNE["rccl_tables"] = 1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X (no CPU fallback in the product path)")
    # test hooks (single-GPU boxes): BENCH_DEVICE pins every rank to one device, BENCH_DIST_BACKEND=gloo avoids RCCL's
    # one-rank-per-GPU rule.  The driver's runs use neither: one rank per GPU over RCCL.
    if "BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE"])
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

        def barrier():
            dist.barrier()
    else:
        def barrier():
            pass

    if args.inproc:
        assert world == 1, "--inproc is the single-process form: run it without torch.distributed.run"
        ngpus = min(args.gpus, torch.cuda.device_count())
        r = run_workload_inproc(args.workload, args.batch, args.steps, args.warmup, ngpus)
        mpix = args.steps * r["out_pixels_per_step"] / r["wall_s"] / 1e6
        ach = r["alg_bytes_per_step"] / (r["kernel_ms_avg"] * 1e-3) / 1e9
        print(json.dumps({"metric": "Mpixels/sec sws_scale 4K yuv420p->rgb24 bicubic" if r["name"].startswith("c2") else f"Mpixels/sec sws_scale {r['name']}",
                          "value": round(mpix, 1), "unit": "Mpixels/s", "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(r["wall_s"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8", "data": "synthetic",
                          "config": {"workload": r["desc"], "frames_per_step_per_gpu": r["batch"], "path": r["path"],
                                     "sharding": f"in-library: one process, one sws_scale_frames() call over frames on {ngpus} GPU(s)", "frames_resident": "HBM"},
                          "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                       "traffic": None, "kernel": r["kernel"], "kernel_ms_avg": round(r["kernel_ms_avg"], 4),
                                       "algorithmic_bytes_per_launch": r["alg_bytes_per_step"], "note": "per-GPU launch set on the home GPU"},
                          "cpu_baseline": None}), flush=True)
        return
    main_res = run_workload(args.workload, args.batch, args.steps, args.warmup, rank, world, local_rank, barrier)
    variants = [] if args.variants in ("", "none") else (AUTO_VARIANTS if args.variants == "auto" and args.workload == "c2a"
                                                          else [] if args.variants == "auto" else args.variants.split(","))
    var_res = []
    for v in variants:
        var_res.append(run_workload(v, 0, args.steps, args.warmup, rank, world, local_rank, barrier))
        if args.variants == "auto" and v in SMALL_BATCH:      # the batch earlier rounds quoted, next to the default one
            r = run_workload(v, SMALL_BATCH[v], args.steps, args.warmup, rank, world, local_rank, barrier)
            r["key"] = f"{v}_x{SMALL_BATCH[v]}"
            var_res.append(r)

    def reduce_max(x):
        if world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def summarize(r):
        wall = reduce_max(r["wall_s"])
        kms = reduce_max(r["kernel_ms_avg"])
        mpix = world * args.steps * r["out_pixels_per_step"] / wall / 1e6
        achieved = r["alg_bytes_per_step"] / (kms * 1e-3) / 1e9
        return wall, kms, mpix, achieved

    wall, kms, mpix, achieved = summarize(main_res)
    out = None
    if rank == 0:
        # HBM bytes per launch from the PMC counters: NOT measured by this run (counters need their own rocprofv3 --pmc passes);
        # read from the committed summary of the latest such passes over the same command and labelled with its source
        traffic, traffic_source, pmc = None, None, {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("_tool_version", 1) < 2:
                    # passes made with the round-5 tool: it averaged the launches of a two-launch workload instead of adding them (half the traffic) -- only the
                    # single-launch workloads' figures stand; the others are dropped until the passes are re-run (tools/pmc_traffic.sh since round 6)
                    pmc = {k: v for k, v in pmc.items() if k.startswith("_") or k in ("c2a", "c2b", "c3a", "c4", "c5")}
                traffic = pmc.get(main_res["name"], {}).get("hbm_bytes_per_launch")
                traffic_source = "profiles/pmc_latest.json (" + str(pmc.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.sh")) + ")"
            except Exception:
                traffic, pmc = None, {}
        out = {
            "metric": "Mpixels/sec sws_scale 4K yuv420p->rgb24 bicubic" if main_res["name"].startswith("c2")
                      else f"Mpixels/sec sws_scale {main_res['name']}",
            "value": round(mpix, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"c3a": "u16", "c3b": "u16 (int32 accumulate)", "c5": "f32->int32"}.get(main_res["name"], "u8"),
            "data": "synthetic",
            "config": {"workload": main_res["desc"], "frames_per_step_per_gpu": main_res["batch"],
                       "parity_pin": PARITY_PIN.get(main_res["name"], "reference goldens"),
                       "path": main_res["path"], "sharding": f"frames x{world} (one rank per GPU, no data-path collective)",
                       "frames_resident": "HBM"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": main_res["kernel"], "kernel_ms_avg": round(kms, 4), "kernel_ms_min": round(main_res["kernel_ms_min"], 4),
                         "kernel_ms_median": round(main_res["kernel_ms_median"], 4),
                         "algorithmic_bytes_per_launch": main_res["alg_bytes_per_step"]},
        }
        if main_res["name"] == "c5":
            out["dtype"] = "f32->int32"
        if var_res:
            out["variants"] = {}
    for r in var_res:
        w2, k2, m2, a2 = summarize(r)
        if rank == 0:
            vt = None      # PMC traffic of the committed passes: only for the batch those passes ran (the workload's default)
            if "key" not in r and r["batch"] == WORKLOADS[r["name"]][8]:
                vt = pmc.get(r["name"], {}).get("hbm_bytes_per_launch")
            out["variants"][r.get("key", r["name"])] = {"workload": r["desc"], "frames_per_step_per_gpu": r["batch"], "value": round(m2, 1), "unit": "Mpixels/s",
                                          "ms_per_step": round(w2 / args.steps * 1e3, 4), "path": r["path"], "kernel": r["kernel"],
                                          "roofline": {"bound": "hbm", "achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                       "frac": round(a2 / HBM_PEAK_GBS, 4), "traffic": vt, "algorithmic_bytes_per_launch": r["alg_bytes_per_step"], "kernel_ms_avg": round(k2, 4),
                                                       "kernel_ms_min": round(r["kernel_ms_min"], 4), "kernel_ms_median": round(r["kernel_ms_median"], 4)}}
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(main_res["name"], args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
