"""Dev tool: straight-line check of s_waitcnt coverage in a gfx9 .s: every VGPR written by a DS / VMEM / SMEM instruction must be
covered by an lgkmcnt / vmcnt wait (in-order counting) before it is read or overwritten.  State is reset at labels (unknown
predecessors), so only same-block violations are found.  usage: check_waitcnt.py kernel.s"""
import re
import sys

lines = [l.rstrip("\n") for l in open(sys.argv[1])]


def regs(tok, pre="v"):
    m = re.fullmatch(pre + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(pre + r"(\d+)", tok)
    return {int(m.group(1))} if m else set()


def toks(t):
    op, _, rest = t.partition(" ")
    return op, [x.strip().split(" ")[0] for x in re.split(r",(?![^\[]*\])", rest) if x.strip()]


lg, vm = [], []          # pending (line, dest vregs)
bad = 0
inasm = False
for n, l in enumerate(lines, 1):
    if "#ASMSTART" in l:
        inasm = True
    if "#ASMEND" in l:
        inasm = False
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        if t.endswith(":"):
            lg, vm = [], []
        continue
    if t.endswith(":"):
        lg, vm = [], []
        continue
    op, tk = toks(t)
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            lg = lg[len(lg) - k:] if k else []
        m = re.search(r"vmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            vm = vm[len(vm) - k:] if k else []
        continue
    allv = set().union(*[regs(x) for x in tk]) if tk else set()
    for name, pend in (("lgkmcnt", lg), ("vmcnt", vm)):
        for pl, pr in pend:
            if pr & allv:
                print(f"line {n}: `{t}` touches v{sorted(pr & allv)} still pending on {name} from line {pl}: {lines[pl - 1].strip()}")
                bad += 1
                break
    if op.startswith("ds_") and not op.startswith("ds_write") and tk:
        lg.append((n, regs(tk[0])))
    elif op.startswith("ds_write"):
        lg.append((n, set()))
    elif op.startswith("s_load") or op.startswith("s_memtime"):
        lg.append((n, set()))
    elif op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
        vm.append((n, set() if " lds" in t else regs(tk[0])))
    elif op.startswith(("global_store", "buffer_store", "scratch_store", "global_atomic")):
        vm.append((n, set()))
print(f"{bad} uncovered uses")
