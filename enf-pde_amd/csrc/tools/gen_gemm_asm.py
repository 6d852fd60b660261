#!/usr/bin/env python3
"""Generates enf_gemm_asm.h: one hand-scheduled bf16 GEMM stage per (KB, MTS) shape.

A stage multiplies MTS out-tiles x KB in-blocks of A fragments (1 KB each, order [mt][blk][lane] in
an LDS ring slot) with the wave's KB activation fragments.  hipcc keeps only one or two ds_read_b128
ahead of the MFMA that consumes them (each MFMA then waits most of an LDS round trip); here NBUF
fragment registers are kept in flight with counted lgkmcnt waits, and consecutive MFMAs go to
different accumulators (k-block outer, out-tile inner) so none waits on its predecessor.
The whole stage is ONE asm statement (cdna_hip_programming.md 5.7, form (i)): loads, waits and
MFMAs inside, scratch fragments as early-clobber outputs, so the compiler never touches a register
whose load is still in flight.
"""
import sys

SHAPES = [(4, 8, 8), (2, 4, 6), (2, 8, 8)]        # (KB, MTS, NBUF)
TAIL_NOPS = 11     # MFMA result -> first compiler VALU reader: wait states inside the string


def _acc_ops(MTS, init):
    """operand constraints of the accumulators for an init mode: 'acc' = read-modify-write (caller initialised),
    'zero' = written by the first MFMA with C = 0, 'bias' = loaded from LDS inside the statement"""
    con = "+v" if init == "acc" else "=&v"
    return [f'[a{m}] "{con}"(acc[{m}])' for m in range(MTS)]


def _bias_reads(MTS):
    return [f"ds_read_b128 %[a{mt}], %[bias] offset:{64 * mt}" for mt in range(MTS)]


def emit(KB, MTS, NBUF, init):
    n = KB * MTS
    order = [(mt, blk) for blk in range(KB) for mt in range(MTS)]    # consecutive MFMAs: different accumulators
    off = lambda mt, blk: (mt * KB + blk) * 1024
    L = []
    L.append("s_waitcnt lgkmcnt(0)")              # SMEM returns out of order: start from a clean counter
    if init == "bias":
        L += _bias_reads(MTS)                      # in-order LDS returns: done before the first fragment is
    for i in range(min(NBUF, n)):
        L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i])}")
    for i, (mt, blk) in enumerate(order):
        inflight = min(i + NBUF, n) - i            # reads issued and not yet consumed, this one included
        L.append(f"s_waitcnt lgkmcnt({inflight - 1})")
        c = "0" if (init == "zero" and blk == 0) else f"%[a{mt}]"
        L.append(f"v_mfma_f32_16x16x32_bf16 %[a{mt}], %[s{i % NBUF}], %[b{blk}], {c}")
        if i + NBUF < n:
            L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i + NBUF])}")
    L.append(f"s_nop {TAIL_NOPS}")
    body = "      ENF_ASM_PRIO_ON\n" + "\n".join(f'      "{x}\\n\\t"' for x in L[:-1]) + "\n      ENF_ASM_PRIO_OFF\n" + f'      "{L[-1]}\\n\\t"'
    outs = ", ".join(_acc_ops(MTS, init) + [f'[s{j}] "=&v"(s{j})' for j in range(NBUF)])
    ins = ", ".join([f'[b{k}] "v"(B[{k}])' for k in range(KB)] + ['[addr] "v"(lds_addr)'] + (['[bias] "v"(bias_addr)'] if init == "bias" else []))
    name = {"acc": "run", "zero": "run_zero", "bias": "run_bias"}[init]
    decl = ", ".join(f"s{j}" for j in range(NBUF))
    extra = ", unsigned bias_addr" if init == "bias" else ""
    return f"""  static __device__ __forceinline__ void {name}(f32x4* acc, const bf16x8* B, unsigned lds_addr{extra}) {{
    f32x4 {decl};
    asm volatile(
{body}
      : {outs}
      : {ins});
  }}
"""


def emit_both(KB, MTS, NBUF, mask, name, init="acc"):
    """Transposed product into acc[mt] AND, for the out-tiles in `mask`, the flipped product (operands swapped:
    rows = the wave's columns) into af[.] from the same fragment read."""
    n = KB * MTS
    order = [(mt, blk) for blk in range(KB) for mt in range(MTS)]
    off = lambda mt, blk: (mt * KB + blk) * 1024
    fl = [mt for mt in range(MTS) if mask >> mt & 1]
    L = ["s_waitcnt lgkmcnt(0)"]
    if init == "bias":
        L += _bias_reads(MTS)
    for i in range(min(NBUF, n)):
        L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i])}")
    for i, (mt, blk) in enumerate(order):
        inflight = min(i + NBUF, n) - i
        L.append(f"s_waitcnt lgkmcnt({inflight - 1})")
        c = "0" if (init == "zero" and blk == 0) else f"%[a{mt}]"
        L.append(f"v_mfma_f32_16x16x32_bf16 %[a{mt}], %[s{i % NBUF}], %[b{blk}], {c}")
        if mt in fl:
            L.append(f"v_mfma_f32_16x16x32_bf16 %[f{fl.index(mt)}], %[b{blk}], %[s{i % NBUF}], %[f{fl.index(mt)}]")
        if i + NBUF < n:
            L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i + NBUF])}")
    L.append(f"s_nop {TAIL_NOPS}")
    body = "      ENF_ASM_PRIO_ON\n" + "\n".join(f'      "{x}\\n\\t"' for x in L[:-1]) + "\n      ENF_ASM_PRIO_OFF\n" + f'      "{L[-1]}\\n\\t"'
    outs = ", ".join(_acc_ops(MTS, init) + [f'[f{j}] "+v"(af[{j}])' for j in range(len(fl))] +
                     [f'[s{j}] "=&v"(s{j})' for j in range(NBUF)])
    ins = ", ".join([f'[b{k}] "v"(B[{k}])' for k in range(KB)] + ['[addr] "v"(lds_addr)'] + (['[bias] "v"(bias_addr)'] if init == "bias" else []))
    decl = ", ".join(f"s{j}" for j in range(NBUF))
    name = name + {"acc": "", "zero": "_zero", "bias": "_bias"}[init]
    extra = ", unsigned bias_addr" if init == "bias" else ""
    return f"""  // af[j] <-> out-tile {fl}
  static constexpr int {name}_nflip = {len(fl)};
  static __device__ __forceinline__ void {name}(f32x4* acc, f32x4* af, const bf16x8* B, unsigned lds_addr{extra}) {{
    f32x4 {decl};
    asm volatile(
{body}
      : {outs}
      : {ins});
  }}
"""


def emit_flip(KB, MTS, NBUF, init):
    """Flipped product only (operands swapped: rows = the wave's columns), one accumulator per out-tile."""
    n = KB * MTS
    order = [(mt, blk) for blk in range(KB) for mt in range(MTS)]
    off = lambda mt, blk: (mt * KB + blk) * 1024
    L = ["s_waitcnt lgkmcnt(0)"]
    for i in range(min(NBUF, n)):
        L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i])}")
    for i, (mt, blk) in enumerate(order):
        inflight = min(i + NBUF, n) - i
        L.append(f"s_waitcnt lgkmcnt({inflight - 1})")
        c = "0" if (init == "zero" and blk == 0) else f"%[a{mt}]"
        L.append(f"v_mfma_f32_16x16x32_bf16 %[a{mt}], %[b{blk}], %[s{i % NBUF}], {c}")
        if i + NBUF < n:
            L.append(f"ds_read_b128 %[s{i % NBUF}], %[addr] offset:{off(*order[i + NBUF])}")
    L.append(f"s_nop {TAIL_NOPS}")
    body = "      ENF_ASM_PRIO_ON\n" + "\n".join(f'      "{x}\\n\\t"' for x in L[:-1]) + "\n      ENF_ASM_PRIO_OFF\n" + f'      "{L[-1]}\\n\\t"'
    outs = ", ".join(_acc_ops(MTS, init) + [f'[s{j}] "=&v"(s{j})' for j in range(NBUF)])
    ins = ", ".join([f'[b{k}] "v"(B[{k}])' for k in range(KB)] + ['[addr] "v"(lds_addr)'])
    name = {"acc": "run_flip", "zero": "run_flip_zero"}[init]
    decl = ", ".join(f"s{j}" for j in range(NBUF))
    return f"""  static __device__ __forceinline__ void {name}(f32x4* acc, const bf16x8* B, unsigned lds_addr) {{
    f32x4 {decl};
    asm volatile(
{body}
      : {outs}
      : {ins});
  }}
"""


BOTH = {(4, 8): [(0xFF, "run_both", 8), (0x33, "run_gb", 8)], (2, 4): [(0xF, "run_both", 6)], (2, 8): [(0x33, "run_gb", 8)]}


def main(path):
    # LITE = true: half the fragment registers in flight (4), for kernels short of registers
    out = ["// GENERATED by tools/gen_gemm_asm.py -- do not edit.", "#pragma once", "",
           "// ENF_ASM_PRIO=1 (A/B builds): s_setprio 1 over a stage's fragment reads and MFMAs, back to 0 before its tail wait states",
           "#ifndef ENF_ASM_PRIO", "#define ENF_ASM_PRIO 0", "#endif", "#if ENF_ASM_PRIO",
           '#define ENF_ASM_PRIO_ON "s_setprio 1\\n\\t"', '#define ENF_ASM_PRIO_OFF "s_setprio 0\\n\\t"', "#else",
           '#define ENF_ASM_PRIO_ON', '#define ENF_ASM_PRIO_OFF', "#endif", "",
           "template <int KB, int MTS, bool LITE = false> struct GemmStageAsm { static constexpr bool available = false; };", ""]
    for lite in (False, True):
        for KB, MTS, NBUF in SHAPES:
            nb0 = min(NBUF, 4) if lite else NBUF
            out.append(f"template <> struct GemmStageAsm<{KB}, {MTS}, {'true' if lite else 'false'}> {{")
            out.append("  static constexpr bool available = true;")
            for init in ("acc", "zero", "bias"):
                out.append(emit(KB, MTS, nb0, init))
            for init in ("acc", "zero"):
                out.append(emit_flip(KB, MTS, nb0, init))
            for mask, name, nb in BOTH.get((KB, MTS), []):
                for init in ("acc", "zero", "bias"):
                    out.append(emit_both(KB, MTS, min(nb, 4) if lite else nb, mask, name, init))
            out.append("};\n")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "enf_gemm_asm.h")
