#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing.

    python tools/isa_blocks.py sym.s 'demod_sym_kernelILi6ELi4EfLi0' [--min N]

Prints, in file order, every basic block with its VALU / SALU / LDS / VMEM / branch / waitcnt counts and marks blocks
that hold an s_barrier; the per-role loops of the wavefront pipelines are the stretches between barriers.
"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_barrier"):
        return "bar"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    blocks = []
    cur = {"name": "entry", "line": start, "n": {}}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "line": i, "n": {}}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        k = classify(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k == "br":
            cur.setdefault("targets", []).append(s.split()[-1])
    blocks.append(cur)
    keys = ["valu", "salu", "lds", "vmem", "smem", "wait", "br", "bar"]
    print("%-12s %6s " % ("block", "line") + " ".join("%5s" % k for k in keys) + "  targets")
    tot = {k: 0 for k in keys}
    for b in blocks:
        for k in keys:
            tot[k] += b["n"].get(k, 0)
        print("%-12s %6d " % (b["name"], b["line"] + 1) + " ".join("%5d" % b["n"].get(k, 0) for k in keys) + "  " + ",".join(b.get("targets", [])))
    print("%-12s %6s " % ("total", "") + " ".join("%5d" % tot[k] for k in keys))


if __name__ == "__main__":
    main()
