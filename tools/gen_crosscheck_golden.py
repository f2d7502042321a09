#!/usr/bin/env python3
"""Regenerates tests/golden/reference_answers_r06.json: MD5s of the REAL reference's output (a C-only build under /tmp: tools/ref_vs_port.sh, tools/ref/ref_batch.c) for the
conversions in which round 6's cross-check found oracle and product differing from the reference -- the 8 / 4 bpp ordered-dither converters on widths with a 4-pixel and a
2-pixel tail, and yuva420p10le / yuva420p16le into p010le / p016le at the same size.  Inputs are tests/oracle_lib.fill_random pictures (seed in the file).  Build container only."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as OL  # noqa: E402

EXE = os.path.join(os.environ.get("REFBUILD", "/tmp/refbuild"), "ref_batch")
BX, POINT, BICUBIC = 0x80000, 0x10, 4
CASES = []
for w in (6, 14, 22, 38, 46, 34, 36, 40, 94):
    for sf, df in (("yuv420p", "rgb4"), ("yuv420p", "rgb8"), ("yuvj420p", "bgr4_byte"), ("yuv422p", "bgr8"), ("yuv420p", "rgb4_byte"), ("yuv422p", "bgr4")):
        CASES.append((w, 20, sf, w, 20, df, BICUBIC | BX, 1000 + w))
for sf in ("yuva420p10le", "yuva420p16le"):
    for df in ("p010le", "p016le"):
        for (w, h) in ((76, 72), (315, 88), (13, 70), (64, 36)):
            CASES.append((w, h, sf, w, h, df, POINT | BX, 2000 + w))
# the same format in the other byte order under SWS_SRC_V_CHR_DROP: bswap_16bpc's row count / the scaler for the unlisted families (round 6, third finding)
for sf, df in (("yuv420p10be", "yuv420p10le"), ("yuv440p10le", "yuv440p10be"), ("yuv444p10be", "yuv444p10le"), ("yuv422p10be", "yuv422p10le"), ("yuva420p10be", "yuva420p10le"),
               ("p010be", "p010le"), ("gbrp10be", "gbrp10le"), ("rgb565be", "rgb565le"), ("gbrpf32be", "gbrpf32le"), ("gray16be", "gray16le")):
    for drop in (0, 1, 2):
        CASES.append((64, 36, sf, 64, 36, df, BICUBIC | BX | (drop << 16), 3000 + drop))

# xyz12 against rgb48 in the other byte order (xyz12 IS rgb48 inside the scaler, utils.c:822-823: bswap_16bpc applies, between the XYZ conversions), and the alpha-blend cascade into
# the alpha-less twin in big-endian order under SWS_SRC_V_CHR_DROP (its second step is native -> big-endian: bswap_16bpc's row count again) -- round 6, fourth finding
for sf, df in (("xyz12le", "rgb48be"), ("xyz12be", "rgb48le"), ("rgb48be", "xyz12le"), ("xyz12le", "xyz12be")):
    CASES.append((64, 36, sf, 64, 36, df, BICUBIC | BX, 4000))
for drop in (0, 1):
    CASES.append((64, 36, "yuva420p10be", 64, 36, "yuv420p10be", 2 | BX | (drop << 16), 4100 + drop, dict(alpha_blend=1)))
    CASES.append((64, 36, "yuva420p10le", 64, 36, "yuv420p10be", 2 | BX | (drop << 16), 4200 + drop, dict(alpha_blend=2)))


def main():
    out = []
    for case in CASES:
        (sw, sh, sf, dw, dh, df, flags, seed), opts = case[:8], (case[8] if len(case) > 8 else {})
        src = OL.fill_random(OL.Frame(sf, sw, sh), seed)
        inp = f"CASE {sw} {sh} {sf} {dw} {dh} {df} {flags} 165 {1 if opts else 0} 1 0 0 -513 -513 -513 -513 0 0 0 0 0 0 0 0 {opts.get('alpha_blend', 0)} 0 123456 123456\n".encode()
        inp += b"".join(np.ascontiguousarray(a[:, :rb]).tobytes() for a, rb in zip(src.planes, src.row_bytes))
        o = subprocess.run([EXE], input=inp, capture_output=True, check=True).stdout
        hdr, data = o[:o.index(b"\n")].split(), o[o.index(b"\n") + 1:]
        assert int(hdr[1]) == dh and len(data) == int(hdr[2])
        out.append({"case": [sw, sh, sf, dw, dh, df, flags], "seed": seed, "prefill": 165, "md5": hashlib.md5(data).hexdigest(), **({"opts": opts} if opts else {})})
    json.dump({"_source": "tools/gen_crosscheck_golden.py: the real reference (C-only build), whole destination pictures (visible rows, planes in order)", "cases": out},
              open(os.path.join(ROOT, "tests", "golden", "reference_answers_r06.json"), "w"), indent=0)
    print(len(out), "cases")


if __name__ == "__main__":
    main()
