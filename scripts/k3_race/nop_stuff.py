"""Dev tool: insert `s_nop N` after every instruction of a line range of one kernel in a device .s (timing/hazard bisect).
usage: nop_stuff.py in.s out.s KERNEL_SYMBOL first last [nop_count] [opcode-regex]   (first/last relative to the kernel's label)"""
import re
import sys

src, dst, sym, first, last = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 3
rx = re.compile(sys.argv[7]) if len(sys.argv) > 7 else None
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
out, k, inasm = [], 0, False
for i, l in enumerate(lines):
    out.append(l)
    if "#ASMSTART" in l:
        inasm = True
    if "#ASMEND" in l:
        inasm = False
    rel = i - start
    t = l.split(";")[0].strip()
    if first <= rel <= last and t and not t.endswith(":") and not t.startswith(".") and not inasm:
        op = t.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            continue
        if rx and not rx.search(op):
            continue
        out.extend([f"\ts_nop {n}"] * int(__import__("os").environ.get("NOP_REPEAT", "1")))
        k += 1
open(dst, "w").write("\n".join(out))
print(f"inserted {k} s_nop {n}")
