#!/usr/bin/env python3
"""tests/hipemu: copies librempeg_amd/csrc into a build directory with the few textual changes an x86 compile of the kernels needs -- the product sources are not
touched and carry no emulation path.  What changes:
  * inline asm: `s_waitcnt ...` -> a wave-level synchronisation point of the emulator, `s_barrier` -> the block barrier, the empty register pins -> nothing,
    `buffer_load_dwordx4 ... lds` (the LDS-DMA request of kernels_strip.hpp) -> hipemu::dma16, the two v_ashr_pk_u8_i32 packs and the v_dot2_i32_i16 forms -> their
    arithmetic (hipemu::pack4_u8_shr / dot2_i16);
  * `extern __shared__` -> an extern array (the emulator's LDS arena), `__shared__` -> a function-local static;
  * the global address-space attribute of wave_util.hpp and the occupancy attribute of the strip kernels go;
  * one exchange the sources do not mark (they need not, on a lockstep wave) gets its synchronisation point: after `stage();` / `sp_stage(P);`.
usage: prep.py <csrc> <out>"""
import os
import re
import shutil
import sys


def find_asm(text, start):
    m = re.compile(r'\basm\s*(volatile\s*)?\(').search(text, start)
    if not m:
        return None
    i, depth, in_str = m.end(), 1, False
    while depth:
        ch = text[i]
        if in_str:
            if ch == '\\':
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
        elif ch == '(':
            depth += 1
        elif ch == ')':
            depth -= 1
        i += 1
    j = i
    while text[j] in ' \t':
        j += 1
    assert text[j] == ';', text[m.start():j + 20]
    return m.start(), j + 1, text[m.end():i - 1]


def replacement(body, where):
    strings, i = "", 0                   # the template: the string literals before the first ':' outside a literal
    while i < len(body) and body[i] != ':':
        if body[i] == '"':
            j = i + 1
            while body[j] != '"':
                j += 2 if body[j] == '\\' else 1
            strings += body[i + 1:j]
            i = j
        i += 1
    if strings == "":
        return "/* (register pin) */ ;"
    if "buffer_load_dwordx4" in strings and " lds" in strings:
        return "(void)keep; hipemu::dma16(lds_dst, voff, rsrc, soff, mask_lo, mask_hi);"
    if "s_barrier" in strings:
        return "hipemu::sync_block();"
    if "s_waitcnt" in strings:
        return "hipemu::sync_wave();"
    m = re.search(r'v_ashr_pk_u8_i32 %0, %1, %2, (\d+)', strings)
    if m:
        return f"d = hipemu::pack4_u8_shr(t0, t1, t2, t3, {m.group(1)});"
    if "v_dot2_i32_i16" in strings:
        ops = re.findall(r'"[vs]"\((\w+)\)', body)
        return f"d = hipemu::dot2_i16({ops[0]}, {ops[1]}, 0);"
    raise SystemExit(f"prep.py: an asm statement this script does not know, in {where}: {strings[:80]}")


def convert(text, where):
    out, pos = [], 0
    while True:
        hit = find_asm(text, pos)
        if not hit:
            break
        a, b, body = hit
        line_start = text.rfind("\n", 0, a) + 1
        if text[line_start:a].lstrip().startswith("//"):        # an asm statement quoted in a comment
            out.append(text[pos:b]); pos = b
            continue
        out.append(text[pos:a]); out.append(replacement(body, where)); pos = b
    out.append(text[pos:])
    text = "".join(out)
    # the register-staged strip kernels fill their LDS rows for the first time (stage()) and read them in the same straight line: in-order LDS access of a lockstep
    # wave on the GPU, an exchange between fibers here -- the emulator needs the point marked
    text = re.sub(r'(?m)^(\s*)stage\(\);', r'\1stage(); hipemu::sync_wave();', text)
    text = re.sub(r'(?m)^(\s*)sp_stage\(P\);', r'\1sp_stage(P); hipemu::sync_wave();', text)
    text = re.sub(r'\bextern\s+__shared__', 'extern', text)
    text = re.sub(r'\b__shared__', 'static', text)
    text = text.replace("__attribute__((address_space(1)))", "")
    text = re.sub(r'__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)', '', text)
    text = text.replace("uint64_t keep;", "uint64_t keep = 0;")
    return text


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for name in sorted(os.listdir(src)):
        p = os.path.join(src, name)
        if not os.path.isfile(p):
            continue
        if name.endswith((".hip", ".hpp", ".cpp", ".h")):
            new = convert(open(p).read(), name)
            q = os.path.join(dst, name)
            if not os.path.exists(q) or open(q).read() != new:
                open(q, "w").write(new)
        elif name == "Makefile":
            shutil.copy(p, os.path.join(dst, name))
        elif name.endswith(".v"):           # the emulator's counters are exported next to the library's own symbols
            open(os.path.join(dst, name), "w").write(open(p).read().replace("global:", "global:\n        hipemu_*;", 1))


if __name__ == "__main__":
    main()
