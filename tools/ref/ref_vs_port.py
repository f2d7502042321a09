#!/usr/bin/env python3
"""second half of tools/ref_vs_port.sh: runs the reference timer and the port on the BASELINE configurations, writes profiles/ref_vs_port.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as OL  # noqa: E402

BICUBIC, BILINEAR, LANCZOS, BITEXACT, ACCURATE = 4, 2, 0x200, 0x80000, 0x40000
CONFIGS = {  # name: srcW srcH srcFmt dstW dstH dstFmt flags bt2020 reps
    "c1": (1280, 720, "yuv420p", 640, 360, "yuv420p", BILINEAR | BITEXACT, 0, 40),
    "c2a": (3840, 2160, "yuv420p", 3840, 2160, "rgb24", BICUBIC | BITEXACT, 0, 12),
    "c2b": (3840, 2160, "yuv420p", 3840, 2160, "rgb24", BICUBIC | BITEXACT | ACCURATE, 0, 6),
    "c3a": (7680, 4320, "yuv420p10le", 7680, 4320, "p010le", LANCZOS | BITEXACT, 0, 6),
    "c3b": (7680, 4320, "yuv420p10le", 3840, 2160, "p010le", LANCZOS | BITEXACT, 0, 4),
    "c4": (1920, 1080, "nv12", 1920, 1080, "bgr0", BICUBIC | BITEXACT, 0, 12),
    "c5": (3840, 2160, "gbrpf32le", 3840, 2160, "yuv444p16le", BICUBIC | BITEXACT, 1, 4),
}


def port_ms(sw, sh, sf, dw, dh, df, flags, bt2020, reps, src):
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    if bt2020:
        o.set_colorspace(9, 1, 9, 1)
    dst = OL.Frame(df, dw, dh)
    o.scale(src, dst)
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        o.scale(src, dst)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def main():
    timer = sys.argv[1]
    out = {"_source": "tools/ref_vs_port.sh in the build container: C-only reference (configure --disable-asm, gcc, /tmp build) vs oracle/ "
                      "(gcc -O3 -fno-tree-vectorize), one thread, best of N; ms per frame",
           "_cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip()}
    for name, cfg in CONFIGS.items():
        sw, sh, sf = cfg[:3]
        src = OL.fill_random(OL.Frame(sf, sw, sh), 77)       # one picture for both sides (clipping / rounding paths are data dependent)
        with open("/tmp/ref_vs_port_src.bin", "wb") as f:
            for pl, rb in zip(src.planes, src.row_bytes):
                f.write(pl[:, :rb].tobytes())
        r = subprocess.check_output([timer, name] + [str(x) for x in cfg] + ["/tmp/ref_vs_port_src.bin"], stderr=subprocess.DEVNULL).decode().split()
        ref = float(r[1])
        port = port_ms(*cfg, src)
        out[name] = {"reference_ms": round(ref, 3), "port_ms": round(port, 3), "port_over_reference": round(port / ref, 3)}
        print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "ref_vs_port.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
