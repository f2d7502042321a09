#!/usr/bin/env python3
"""Collect tests/golden/fate_nut_md5.json from the reference tree (build container only: needs /root/reference).

  * the known answers of fate-filter-pixfmts-copy / -null / -scale (tests/ref/fate/filter-pixfmts-*: "<pix_fmt> <md5>") and of
    fate-filter-pixdesc-<pix_fmt> (tests/ref/fate/filter-pixdesc-*): MD5s of the NUT files `ffmpeg ... -vcodec rawvideo -f nut md5:` writes
    (tests/fate-run.sh:621-660, tests/fate/filter-video.mak:653-713);
  * per pixel format the fourcc the NUT stream header carries: rawenc.c:41-43 takes the FIRST entry of libavcodec/raw_pix_fmt_tags.h for the
    format, libavformat/mux.c:304-325 keeps it (it has to sit among the muxer's RAWVIDEO tags: nut.c ff_nut_video_tags, riff.c ff_codec_bmp_tags)
    or, for a format without an entry, takes the muxer's first RAWVIDEO tag (nut.c:50).
Data only: tables of names, tags and checksums."""
import json, os, re, sys
REF = os.environ.get("SWS_REFERENCE_ROOT", sys.argv[1] if len(sys.argv) > 1 else "/root/reference")   # reference tree: argv[1] or $SWS_REFERENCE_ROOT
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mktag(args):
    v = []
    for a in args:
        a = a.strip()
        v.append(ord(a[1]) if a.startswith("'") else int(a, 0))
    return v[0] | v[1] << 8 | v[2] << 16 | v[3] << 24


# AV_PIX_FMT_* -> pixdesc name
names = {}
for m in re.finditer(r"\[AV_PIX_FMT_([A-Z0-9_]+)\]\s*=\s*\{\s*\.name\s*=\s*\"([a-z0-9_]+)\"", open(f"{REF}/libavutil/pixdesc.c").read()):
    names[m.group(1)] = m.group(2)
# native-endian aliases (a little-endian host)
for m in re.finditer(r"#define\s+AV_PIX_FMT_([A-Z0-9_]+)\s+AV_PIX_FMT_NE\(\s*([A-Z0-9_]+)\s*,\s*([A-Z0-9_]+)\s*\)", open(f"{REF}/libavutil/pixfmt.h").read()):
    if m.group(3) in names:
        names[m.group(1)] = names[m.group(3)]

raw_first = {}
for m in re.finditer(r"\{\s*AV_PIX_FMT_([A-Z0-9_]+)\s*,\s*MKTAG\(([^)]*)\)\s*\}", open(f"{REF}/libavcodec/raw_pix_fmt_tags.h").read()):
    n = names.get(m.group(1))
    if n and n not in raw_first:
        raw_first[n] = mktag(m.group(2).split(","))

mux_raw = []
for fn in ("libavformat/nut.c", "libavformat/riff.c"):
    for m in re.finditer(r"\{\s*AV_CODEC_ID_RAWVIDEO\s*,\s*MKTAG\(([^)]*)\)\s*\}", open(f"{REF}/{fn}").read()):
        mux_raw.append(mktag(m.group(1).split(",")))
up = lambda t: int.from_bytes(bytes(c - 32 if 97 <= c <= 122 else c for c in t.to_bytes(4, "little")), "little")
mux_up = {up(t) for t in mux_raw}

out = {"_source": "tools/gen_fate_nut_golden.py over /root/reference (tests/ref/fate/filter-pixfmts-{copy,null,scale}, filter-pixdesc-*; raw_pix_fmt_tags.h, nut.c, riff.c)"}
for t in ("copy", "null", "scale"):
    out[t] = {l.strip()[:-32].strip(): l.strip()[-32:] for l in open(f"{REF}/tests/ref/fate/filter-pixfmts-{t}") if l.strip()}      # printf '%-20s' label, then the MD5
out["pixdesc"] = {}
for fn in sorted(os.listdir(f"{REF}/tests/ref/fate")):
    if fn.startswith("filter-pixdesc-"):
        l = open(f"{REF}/tests/ref/fate/{fn}").read().strip()
        out["pixdesc"][l[:-32].strip()[len("pixdesc-"):]] = l[-32:]
tags = {}
for f in sorted(set(out["copy"]) | set(out["scale"]) | set(out["pixdesc"])):
    t = raw_first.get(f, 0)
    if t and up(t) not in mux_up:
        print("warning:", f, "raw tag", t.to_bytes(4, "little"), "is not a RAWVIDEO tag of the NUT muxer", file=sys.stderr)
    tags[f] = t if t else mux_raw[0]
out["tags"] = tags
dst = os.path.join(root, "tests", "golden", "fate_nut_md5.json")
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print("wrote", dst, {k: len(v) for k, v in out.items() if isinstance(v, dict)}, "formats without a raw tag:", sorted(f for f in tags if f not in raw_first))
