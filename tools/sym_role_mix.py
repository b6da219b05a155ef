#!/usr/bin/env python3
"""Round 6: static instruction mix of every role's step in the symbol-paced kernel -- compiles same_kernels_sym.hip with
-DSYM_ASM_MARKS (assembly comments where a role's step begins and ends) and counts, per role, the instructions between the marks in
listing order (rare regions included: a yardstick for "did this change remove instructions", not a dynamic count).
    python tools/sym_role_mix.py [kernel substring, default the 22.05 kHz time-major build]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sameold_amd import build as b
want = sys.argv[1] if len(sys.argv) > 1 else "demod_sym_kernelILi42ELi6ELi4EfLi0"
out = os.path.join(tempfile.gettempdir(), "sym_marks.s")
# (the 44.1 / 48 kHz instantiations are a translation unit of their own, with flags of their own: sameold_amd/build.py)
src = "same_kernels_sym.hip" if "ILi42E" in want else "same_kernels_sym_hi.hip"
cmd = [b.hipcc()] + b.flags() + b.SOURCE_FLAGS.get(src, []) + ["-DSYM_ASM_MARKS", "--cuda-device-only", "-S", os.path.join(b.CSRC, src), "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
names = ["S", "T", "A", "E", "Y1", "Y2"]
inside, cur, counts = False, None, {}
for line in open(out):
    if line.startswith("_ZN4same16demod_sym_kernel"):
        inside = want in line
        continue
    if not inside:
        continue
    m = re.search(r"; SYMMARK (\d) (\d)", line)
    if m:
        r, k = int(m.group(1)), int(m.group(2))
        if k == 0: cur = r; counts.setdefault(r, {})
        elif k == 2 and cur == r: cur = None
        continue
    if cur is None:
        continue
    t = line.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("mov" if op.startswith("v_mov") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_sleep"))
            else "wait" if op.startswith(("s_waitcnt", "s_nop", "s_sleep")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else "lds" if op.startswith("ds_") else "vmem")
    counts[cur][kind] = counts[cur].get(kind, 0) + 1
print(f"{want}: instructions between a role's step marks, listing order")
for r in sorted(counts):
    c = counts[r]
    print(f"  {names[r]:3s} total {sum(c.values()):5d}  " + "  ".join(f"{k} {c.get(k, 0)}" for k in ("valu", "mov", "salu", "lds", "vmem", "wait", "branch")))
