#!/usr/bin/env python3
"""Instruction census of the loops of a gfx950 kernel from `hipcc -S --cuda-device-only` output.

usage: isa_count.py file.s kernel-substring [min_instructions]

Finds every backward branch (label defined above its use), and prints for the span label..branch the instruction
mix: fp64 VALU, other VALU, transcendental, SALU, LDS, global/flat, waitcnt.  The kernels of this repository are
bound by instruction issue, so these counts are what a change has to move; the script lets
that be checked on the build box, without a GPU.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
            return "trans"
        if "f64" in op or op in ("v_lshl_add_u64", "v_mad_u64_u32"):
            return "v64"
        return "v32"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    # kernel extents
    start = None
    for n, l in enumerate(lines):
        if l.startswith("_Z") and l.split(":")[0].find(key) >= 0 and l.rstrip().endswith(("E", ")")) is False:
            pass
        m = re.match(r"^(_Z\S+):", l)
        if m and key in m.group(1):
            start = n
            name = m.group(1)
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while end < len(lines) and not lines[end].startswith("\t.section") and ".end_amdhsa_kernel" not in lines[end] and not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    print("kernel", name, "lines", len(body))
    labels = {}
    ins = []  # (index in body, op, text)
    for n, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        if re.match(r"^[a-z]", op):
            ins.append((n, op, t))
    total = Counter(classify(op) for _, op, _ in ins)
    print("whole kernel:", dict(total), "total", len(ins))
    loops = []
    for k, (n, op, t) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    for a, b, tgt in sorted(loops):
        if b - a < min_ins:
            continue
        c = Counter(classify(op) for _, op, _ in ins[a:b + 1])
        ops = Counter(op for _, op, _ in ins[a:b + 1])
        print("\nloop %s: %d instructions  %s" % (tgt, b - a + 1, dict(c)))
        print("   ", ", ".join("%s x%d" % kv for kv in ops.most_common(28)))


if __name__ == "__main__":
    main()
