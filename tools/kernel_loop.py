"""Print (or summarise) the hot loop of one kernel from a device assembly dump.

  python tools/kernel_loop.py build/flash_hip.s <mangled-name-substring> [--dump]

The hot loop is the innermost loop (a label that a later s_cbranch jumps back to) that holds the most
MFMAs.  Without --dump: instruction mix of the loop and, per MFMA, the instructions issued behind it.
"""
import collections
import re
import sys


def kernel_body(lines, name):
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^[A-Za-z_0-9]+:", l) and name in l and not l.startswith(".L"):
            start = i
        elif start is not None and l.strip().startswith("s_endpgm"):
            # a kernel may hold several s_endpgm (early exits); stop at the .Lfunc_end label
            pass
        elif start is not None and l.startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit(f"kernel {name} not found")


def hot_loop(body):
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for j, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            loops.append((labels[m.group(1)], j))
    best = None
    for (i, j) in loops:
        if any((a, b) != (i, j) and i <= a and b <= j for (a, b) in loops):
            continue  # not innermost
        n = sum("v_mfma" in x for x in body[i:j])
        if best is None or n > best[0]:
            best = (n, i, j)
    return body[best[1]:best[2] + 1]


def main():
    src, name = sys.argv[1], sys.argv[2]
    lines = [l.rstrip("\n") for l in open(src)]
    loop = [l.strip() for l in hot_loop(kernel_body(lines, name)) if l.strip() and not l.strip().startswith(";")]
    if "--dump" in sys.argv:
        print("\n".join(loop))
        return
    ops = collections.Counter(l.split()[0] for l in loop if not l.startswith(".L"))
    print("loop lines", len(loop), "mfma", sum(v for k, v in ops.items() if k.startswith("v_mfma")))
    print(ops.most_common(40))


if __name__ == "__main__":
    main()
