#!/usr/bin/env python3
"""Regenerate include/swscale_hip_prefix.h from the dynamic symbol table of librempeg_amd/lib/libswscale_hip.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "librempeg_amd", "lib", "libswscale_hip.so")], text=True)
syms = sorted(l.split()[2].split("@")[0] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")
ren = lambda s: "swship_" + s[4:] if s.startswith("sws_") else "swship_" + s
head = open(os.path.join(ROOT, "include", "swscale_hip_prefix.h")).read().split("#ifndef SWSCALE_HIP_PREFIX_H")[0]
body = ["#ifndef SWSCALE_HIP_PREFIX_H", "#define SWSCALE_HIP_PREFIX_H", ""] + [f"#define {s} {ren(s)}" for s in syms] + ["", "#endif /* SWSCALE_HIP_PREFIX_H */", ""]
open(os.path.join(ROOT, "include", "swscale_hip_prefix.h"), "w").write(head + "\n".join(body))
print(len(syms), "symbols")
