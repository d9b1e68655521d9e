#!/usr/bin/env python3
"""Audit of the hand-issued matrix instructions in a gfx950 assembly dump (the compiler pads nothing for
MFMAs issued from inline asm), in both directions:
  * MFMAs whose source registers are written by one of the few instructions right before them;
  * MFMA results that an instruction other than a chained MFMA touches before the result latency is over
    (followed along every path: fall-through, branch targets, loop back edges);
  * buffer loads / stores whose scalar operands a vector instruction wrote fewer than five wait states earlier.

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


# Wait states a matrix instruction's result needs before anything but another matrix instruction's
# accumulator input may touch it (CDNA3/CDNA4 ISA, "XDL write VGPR -> VALU read/write, memory/LDS read":
# passes + 3 on gfx950), by the pass count of the shapes the kernels use.
RESULT_WAIT = {"v_mfma_f32_16x16x32": 8 + 3, "v_mfma_f32_32x32x16": 16 + 3, "v_mfma_f32_4x4x4": 2 + 3}


def result_hazards(lines, follow=40):
    """MFMA destination -> first later instruction that is not an MFMA and names one of its registers:
    list of (mfma, reader, wait states between, wait states required) where too few lie between.
    Counted as the hardware's sequencer does: every instruction is one wait state, `s_nop N` is N + 1;
    an MFMA in between counts one (it occupies more, so this is the pessimistic reading), and so does a
    branch.  The scan follows every path the program can take: through fall-through labels, into the
    target of an unconditional branch, and both ways at a conditional branch (so a matrix instruction at
    the bottom of a loop is checked against the readers at the loop's header, across the back edge, and
    against the out-of-line blocks it can branch into), each path with the wait states collected so far.
    A path ends when the latency is covered, after `follow` instructions, or at the end of the kernel."""
    ins = [l for l in lines]
    label_pos = {}
    for i, l in enumerate(ins):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_pos[m.group(1)] = i
    found = []
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        need = next((w for k, w in RESULT_WAIT.items() if l.startswith(k)), None)
        if need is None:
            continue
        dst = reg_set(l.split(None, 1)[1].split(",")[0])
        work = [(i + 1, 0, 0)]
        visited = set()
        hit = None
        while work and hit is None:
            pos, waited, seen = work.pop()
            while pos < len(ins):
                p = ins[pos]
                pos += 1
                if p.startswith(".LBB"):
                    continue
                if p.startswith(("s_endpgm", "s_setpc")):
                    break
                if p.startswith("s_nop"):
                    waited += int(p.split()[1]) + 1
                    continue
                if waited >= need or seen >= follow:
                    break
                br = re.match(r"s_(branch|cbranch_\w+)\s+(\.LBB\d+_\d+)", p)
                if br:
                    tgt = label_pos.get(br.group(2))
                    if tgt is not None and (tgt, waited) not in visited:
                        visited.add((tgt, waited))
                        work.append((tgt, waited + 1, seen + 1))
                    if br.group(1) == "branch":
                        break
                    waited += 1
                    seen += 1
                    continue
                seen += 1
                touched = reg_set(p.split(None, 1)[1]) if " " in p else set()
                if touched & dst:
                    if p.startswith("v_mfma"):
                        # accumulator chaining (srcC / vDst of the next matrix instruction) is interlocked;
                        # a result used as an A or B operand is not
                        ops = p.split(None, 1)[1].split(",")
                        ab = set()
                        for o in ops[1:3]:
                            ab |= reg_set(o)
                        if ab & dst:
                            hit = (l, p, waited, need)
                        break
                    hit = (l, p, waited, need)
                    break
                waited += 1
        if hit is not None:
            found.append(hit)
    return found


# A vector memory instruction that reads a scalar register which a VECTOR instruction wrote
# (v_readfirstlane, v_readlane, a compare that writes a mask) needs five wait states in between.  The
# compiler pads its own loads; the loads the kernels issue from inline asm are invisible to it.
SGPR_FROM_VALU_WAIT = 5


def sgpr_set(tok):
    out = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", tok):
        out |= set(range(int(lo), int(hi) + 1))
    out |= {int(i) for i in re.findall(r"\bs(\d+)\b", tok)}
    return out


def scalar_operand_hazards(lines):
    """(writer, memory instruction, wait states between) for every buffer load / store one of whose scalar
    operands (descriptor, scalar offset) was written by a vector instruction fewer than five wait states
    before it, looking back along the text (labels are passed through: the pessimistic reading).  A
    scalar instruction in between that rewrites such a register from other scalar registers passes the
    question on to its sources (s_lshl_b32 s8, s8, 1 does not make a fresh s8 old)."""
    ins = [l for l in lines if not l.startswith(".LBB")]
    found = []
    for i, l in enumerate(ins):
        if not l.startswith(("buffer_load", "buffer_store")):
            continue
        used = sgpr_set(l.split(None, 1)[1])
        waited = 0
        j = i - 1
        while j >= 0 and waited < SGPR_FROM_VALU_WAIT:
            p = ins[j]
            j -= 1
            if p.startswith("s_nop"):
                waited += int(p.split()[1]) + 1
                continue
            if " " in p:
                ops = p.split(None, 1)[1].split(",")
                written = sgpr_set(ops[0])
                if p.startswith(("v_readfirstlane", "v_readlane", "v_cmp")):
                    if written & used:
                        found.append((p, l, waited))
                        break
                elif p.startswith("s_") and written & used and not p.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier")):
                    used |= sgpr_set(",".join(ops[1:]))
            waited += 1
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
        r = result_hazards(lines)
        print("   results read too early:", len(r))
        for m, p, w, need in r[:12]:
            print("   ", m, " -> ", p, f"  ({w} of {need} wait states)")
        sh = scalar_operand_hazards(lines)
        print("   memory instructions reading a scalar register a vector instruction just wrote:", len(sh))
        for p, m, w in sh[:12]:
            print("   ", p, " -> ", m, f"  ({w} of {SGPR_FROM_VALU_WAIT} wait states)")
