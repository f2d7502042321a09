#!/usr/bin/env python3
"""Unique sanitizer report sites from the log files tools/asan_env.sh leaves (ASAN_LOG prefix): one line per
(file:line:col, message kind) with a hit count, ASan errors first.  usage: asan_summary.py /tmp/asan_report [more prefixes]"""
import collections
import glob
import re
import sys

ub = collections.Counter()
asan = []
for prefix in sys.argv[1:]:
    for path in sorted(glob.glob(prefix + "*")):
        text = open(path, errors="replace").read()
        for m in re.finditer(r"^(\S+?:\d+:\d+): runtime error: (.*)$", text, re.M):
            kind = re.sub(r"-?\d[\dxa-fA-F]*", "N", m.group(2))
            ub[(m.group(1), kind)] += 1
        for m in re.finditer(r"==\d+==ERROR: AddressSanitizer: (.*)", text):
            # the first frames inside the repo name the culprit
            tail = text[m.end():m.end() + 6000]
            frames = re.findall(r"#\d+ \S+ in (\S+) (/root/repo/\S+|\S*librempeg\S*|\S*oracle\S*)", tail)[:4]
            asan.append((m.group(1)[:120], frames, path))
print(f"AddressSanitizer errors: {len(asan)}")
for what, frames, path in asan:
    print(f"  {what}   [{path}]")
    for fn, where in frames:
        print(f"      {fn}  {where}")
print(f"UndefinedBehaviorSanitizer sites: {len(ub)}")
for (where, kind), n in sorted(ub.items()):
    print(f"  {n:8d}  {where}  {kind}")
