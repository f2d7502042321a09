#!/usr/bin/env python3
"""Writes tests/golden/legacy_format_entries.json: the reference's table of supported input / output pixel formats
(libswscale/format.c legacy_format_entries: { is_supported_in, is_supported_out } per AVPixelFormat), with the enum values of
libavutil/pixfmt.h.  Data only; run here, where /root/reference exists."""
import sys
import json, os, re
REF = os.environ.get("SWS_REFERENCE_ROOT", sys.argv[1] if len(sys.argv) > 1 else "/root/reference")   # reference tree: argv[1] or $SWS_REFERENCE_ROOT
src = open(os.path.join(REF, "libswscale/format.c")).read()
i = src.index("legacy_format_entries"); j = src.index("};", i)
ents = re.findall(r"\[AV_PIX_FMT_(\w+)\]\s*=\s*\{\s*(\d)\s*,\s*(\d)", src[i:j])
pf = open(os.path.join(REF, "libavutil/pixfmt.h")).read()
body = pf[pf.index("enum AVPixelFormat {"):]; body = body[:body.index("AV_PIX_FMT_NB")]
vals, v = {}, -1
for line in body.split("\n"):
    m = re.match(r"\s*AV_PIX_FMT_(\w+)\s*(=\s*(-?\d+))?\s*,", line)
    if m:
        v = int(m.group(3)) if m.group(3) is not None else v + 1
        vals[m.group(1)] = v
nb = v + 1
out = {"nb": nb, "entries": [{"name": n.lower(), "value": vals[n], "in": int(a), "out": int(b)} for n, a, b in ents]}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "legacy_format_entries.json")
json.dump(out, open(path, "w"), indent=0)
print(len(ents), "entries ->", path)
