#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from the gfx950 disassembly of an object file (no GPU): per loop (a backward branch and the code between
its target and itself) the number of VALU / SALU / SMEM / LDS / VMEM / branch / waitcnt instructions, with v_dot2 and v_mov counted apart.
usage: tools/isa_mix.py <obj.o> <kernel name regex> [min loop length]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin/"


def disasm(obj):
    t = tempfile.mkdtemp()
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={t}/f.bin", obj], stdin=subprocess.DEVNULL)
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={t}/f.bin", f"--output={t}/d.co", "--unbundle"],
                          stdin=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return subprocess.check_output([LLVM + "llvm-objdump", "-d", f"{t}/d.co"], stdin=subprocess.DEVNULL, text=True)


def klass(op):
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")): return "branch"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    obj, pat = sys.argv[1], re.compile(sys.argv[2])
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cur, ins = None, {}
    for l in disasm(obj).splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", l)
        if m:
            cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().split("(")[0]
            ins[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if cur and m:
            ins[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    for name, L in ins.items():
        if not pat.search(name) or not L:
            continue
        addr = {a: i for i, (a, _, _) in enumerate(L)}
        loops = []
        for i, (a, op, args) in enumerate(L):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.match(r"(\d+)", args.strip())
                if m:
                    off = int(m.group(1))
                    if off >= 32768:
                        off -= 65536
                    tgt = a + 4 + 4 * off
                    if tgt <= a and tgt in addr and i - addr[tgt] >= minlen:
                        loops.append((addr[tgt], i))
        print(f"## {name}: {len(L)} instructions, {len(loops)} loops of >= {minlen}")
        for lo, hi in sorted(loops):
            c = {}
            for a, op, args in L[lo:hi + 1]:
                k = klass(op)
                c[k] = c.get(k, 0) + 1
                if op.startswith("v_dot2"): c["v_dot2"] = c.get("v_dot2", 0) + 1
                if op.startswith("v_mov_b32"): c["v_mov"] = c.get("v_mov", 0) + 1
                if op.startswith("ds_read") or op.startswith("ds_load"): c["ds_read"] = c.get("ds_read", 0) + 1
            inner = sum(1 for (l2, h2) in loops if l2 > lo and h2 < hi)
            print(f"  loop @{lo}..{hi} ({hi - lo + 1} instr, {inner} nested): " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
