"""Dev tool: linear scan of a gfx950 .s for software-managed MFMA hazards the assembler / hipcc do not pad inside or around
inline asm.  For every v_mfma it looks ahead (straight-line, ignoring control flow) and reports the first VALU / DS / VMEM
instruction that READS the result or WRITES a source of the MFMA within `window` wait states (s_nop N = N + 1 states, every
other instruction 1).  Measured on MI355X (scripts/ubench/mfma_hazard.hip): a VALU read of a v_mfma_f32_16x16x32_bf16 result
is stale with fewer than 7 wait states.
Second rule: a VALU instruction INSIDE an inline-asm block (;;#ASMSTART .. ;;#ASMEND -- invisible to hipcc's hazard
recognizer) whose result is read by a v_mfma as A / B / C fewer than 2 wait states later (scripts/ubench/valu_mfma_hazard.hip:
stale below 2), and a transcendental inside asm whose result any VALU reads with no wait state (trans_hazard.hip: 1 needed).
usage: check_mfma_hazards.py kernel.s [window]"""
import re
import sys

path = sys.argv[1]
window = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lines = [l.rstrip("\n") for l in open(path)]
ins = []
in_asm_line = set()        # line numbers of instructions that sit inside inline-asm blocks
inasm = False
for n, l in enumerate(lines, 1):
    if "#ASMSTART" in l:
        inasm = True
    if "#ASMEND" in l:
        inasm = False
    t = l.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith(".") or t.startswith(";"):
        continue
    ins.append((n, t))
    if inasm:
        in_asm_line.add(n)


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def operands(t):
    op, _, rest = t.partition(" ")
    toks = [x.strip() for x in re.split(r",(?![^\[]*\])", rest) if x.strip()]
    toks = [re.sub(r"\s+(offset|op_sel|neg_|offen|lds|off\b|sc\d|nt).*", "", x) for x in toks]
    return op, toks


def defs_uses(t):
    op, toks = operands(t)
    if not toks:
        return op, set(), set()
    if op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "global_atomic", "ds_bpermute") ) and not op.startswith("ds_bpermute"):
        return op, set(), set().union(*[regs(x) for x in toks])
    d = regs(toks[0])
    u = set().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
    return op, d, u


bad = 0
for i, (n, t) in enumerate(ins):
    if not t.startswith("v_mfma"):
        continue
    op, toks = operands(t)
    D, A, B, C = regs(toks[0]), regs(toks[1]), regs(toks[2]), regs(toks[3]) if len(toks) > 3 else set()
    ws = 0
    for j in range(i + 1, min(i + 40, len(ins))):
        n2, t2 = ins[j]
        if ws >= window:
            break
        op2, d2, u2 = defs_uses(t2)
        if op2.startswith("s_nop"):
            ws += int(t2.split()[1]) + 1
            continue
        if op2.startswith("v_mfma"):
            o2, k2 = operands(t2)
            a2, b2, c2 = regs(k2[1]), regs(k2[2]), regs(k2[3]) if len(k2) > 3 else set()
            if (a2 | b2) & D:
                print(f"{path}:{n2}: MFMA reads result of MFMA at line {n} as A/B after {ws} wait states: {t2}")
                bad += 1
                break
            if c2 & D and c2 != D:
                print(f"{path}:{n2}: MFMA reads a PARTIALLY overlapping result of line {n} as C after {ws}: {t2}")
                bad += 1
                break
        elif u2 & D:
            print(f"{path}:{n2}: reads MFMA result (line {n}: {t}) after {ws} wait states: {t2}")
            bad += 1
            break
        elif op2.startswith("v_") and d2 & (C - D):
            if ws < 3:
                print(f"{path}:{n2}: VALU overwrites srcC of MFMA at line {n} after {ws} wait states: {t2}")
                bad += 1
            break
        ws += 1
# ---- rule 2: asm VALU / trans producers the compiler cannot see
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
for i, (n, t) in enumerate(ins):
    if n not in in_asm_line or not t.startswith("v_") or t.startswith("v_mfma"):
        continue
    op, d, _ = defs_uses(t)
    if not d:
        continue
    need_valu = 1 if op.startswith(TRANS) else 0
    ws = 0
    for j in range(i + 1, min(i + 6, len(ins))):
        n2, t2 = ins[j]
        op2, d2, u2 = defs_uses(t2)
        if op2.startswith("s_nop"):
            ws += int(t2.split()[1]) + 1
            continue
        if op2.startswith("v_mfma"):
            o2, k2 = operands(t2)
            srcs = set().union(*[regs(x) for x in k2[1:]])
            if srcs & d and ws < 2:
                print(f"{path}:{n2}: MFMA reads the result of asm VALU at line {n} ({t}) after {ws} wait states: {t2}")
                bad += 1
                break
        elif op2.startswith("v_") and u2 & d and ws < need_valu:
            print(f"{path}:{n2}: VALU reads the result of asm transcendental at line {n} after {ws} wait states: {t2}")
            bad += 1
            break
        if d2 & d:
            break
        ws += 1
        if ws >= 2:
            break
print(f"{bad} potential hazards (window {window})")
