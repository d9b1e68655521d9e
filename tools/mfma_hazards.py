#!/usr/bin/env python3
"""List MFMAs in a gfx950 assembly dump whose source registers are written by one of the few
instructions right before them (the compiler pads nothing for MFMAs issued from inline asm).

    python tools/mfma_hazards.py build/flash_hip.s [kernel-name-fragment] [window]
"""
import re
import sys


def reg_set(tok):
    """'v[16:19]' / 'a5' / 'v7' -> set of ('v'|'a', index)"""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(kind, i) for i in range(int(lo), int(hi) + 1)}
    for kind, idx in re.findall(r"\b([va])(\d+)\b", tok):
        out.add((kind, int(idx)))
    return out


def hazards(lines, window=2):
    ins = [l for l in lines if not l.startswith(".LBB")]
    found = []
    for i, l in enumerate(ins):
        if "v_mfma" not in l:
            continue
        ops = l.split(None, 1)[1].split(",")
        srcs = set()
        for o in ops[1:]:
            srcs |= reg_set(o)
        pad = 0
        j = i - 1
        seen = 0
        while j >= 0 and seen < window:
            p = ins[j]
            if p.startswith("s_nop"):
                pad += int(p.split()[1]) + 1
                j -= 1
                continue
            seen += 1
            if p.startswith(("v_", "ds_read", "buffer_load", "global_load")) and "v_mfma" not in p and not p.startswith("v_cmp"):
                dst = reg_set(p.split(None, 1)[1].split(",")[0]) if " " in p else set()
                if dst & srcs and pad < 2:
                    found.append((p, l))
                    break
            j -= 1
    return found


def kernels(asm_text):
    out = {}
    for m in re.finditer(r"^(_ZN9flash_hip\S*attn_fwd_kernel[^:\s]*):[^\n]*\n(.*?)^\s*s_endpgm", asm_text, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n")]
        out[m.group(1)] = [l for l in body if l and not l.startswith(";") and (not l.startswith(".") or l.startswith(".LBB"))]
    return out


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    frag = sys.argv[2] if len(sys.argv) > 2 else ""
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for name, lines in kernels(text).items():
        if frag not in name:
            continue
        h = hazards(lines, window)
        print(name, ":", len(h), "suspicious MFMAs of", sum("v_mfma" in l for l in lines))
        for p, l in h[:12]:
            print("   ", p, " -> ", l)
