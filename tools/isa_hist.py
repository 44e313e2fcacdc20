#!/usr/bin/env python3
"""Instruction histogram of one kernel from hipcc's assembly (-S --cuda-device-only), loop by loop.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o k.s lasgun_amd/csrc/k_wavefront.hip
    python tools/isa_hist.py k.s 'wf_trace_kernelILb0ELb1ELb1ELb0ELb0' [--loops] [--min=20]

A "loop" is the label range [target, branch] of a backward branch; loops are listed innermost first with the instruction classes of
their bodies (nested loops' instructions count for the outer loop too).  Classes: f64 = v_add/mul/fma/min/max/... _f64 and the f64
transcendental / division helpers; cmp = v_cmp*; sel = v_cndmask; mov = v_mov / v_accvgpr / v_readlane...; cvt = v_cvt*; int = the
other VALU instructions (integer / logic / address); salu, smem, lds, vmem (global / buffer), scratch, flat.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "sel"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap", "v_perm", "v_bfi")):
            return "mov"
        if op.startswith("v_cvt"):
            return "cvt"
        if "_f64" in op:
            return "f64"
        if "_f32" in op or "_f16" in op:
            return "f32"
        return "int"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load", "s_store")):
            return "smem"
        if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
            return "wait"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("flat_"):
        return "flat"
    if op.startswith(("global_", "buffer_")):
        return "vmem"
    return "other"


def parse(path, kernel):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\S*:", l) and kernel in l:
            start = i
        elif start is not None and l.startswith("\t.size") and kernel in l:
            end = i
            break
    if start is None or end is None:
        raise SystemExit("kernel %r not found in %s" % (kernel, path))
    insts = []  # (index, op, text)
    labels = {}
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.LBB\S*:", s):
            continue
        m = re.match(r"^(\.LBB\S*|\S+):", s)
        if m and not s.startswith(("s_", "v_", "ds_", "global_", "scratch_", "flat_", "buffer_")):
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        if re.match(r"^[a-z]", op):
            insts.append((len(insts), op, s))
    return insts, labels


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) < 2:
        raise SystemExit(__doc__)
    want_loops = "--loops" in sys.argv
    minlen = 20
    for a in sys.argv[1:]:
        if a.startswith("--min="):
            minlen = int(a.split("=")[1])
    insts, labels = parse(args[0], args[1])
    total = collections.Counter(classify(op) for _, op, _ in insts)
    order = ["f64", "cmp", "sel", "mov", "cvt", "int", "f32", "salu", "branch", "wait", "smem", "lds", "vmem", "scratch", "flat", "other"]
    valu = sum(total[c] for c in ("f64", "cmp", "sel", "mov", "cvt", "int", "f32"))
    print("kernel %s: %d instructions, %d VALU (static counts)" % (args[1], len(insts), valu))
    print("  " + "  ".join("%s %d" % (c, total[c]) for c in order if total[c]))
    if not want_loops:
        return
    loops = []
    for i, op, text in insts:
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = text.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    loops.sort(key=lambda x: (x[1] - x[0], x[0]))
    seen = set()
    for a, b, tgt in loops:
        if (a, b) in seen or b - a + 1 < minlen:
            continue
        seen.add((a, b))
        c = collections.Counter(classify(op) for _, op, _ in insts[a:b + 1])
        v = sum(c[k] for k in ("f64", "cmp", "sel", "mov", "cvt", "int", "f32"))
        inner = [(x, y) for x, y, _ in loops if a <= x and y <= b and (x, y) != (a, b)]
        print("loop %-12s insts [%5d, %5d] len %5d  VALU %5d | %s%s" % (tgt, a, b, b - a + 1, v, "  ".join("%s %d" % (k, c[k]) for k in order if c[k]),
                                                                      "  (contains %d inner loops)" % len(set(inner)) if inner else ""))


if __name__ == "__main__":
    main()
