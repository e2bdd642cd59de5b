#!/usr/bin/env python3
"""Hazard check the compiler cannot do: the LDS-DMA of K4f / K7f is an inline-asm `global_load_lds_dwordx4 v, s[a:b]`, and the
hazard recognizer does not look inside inline asm.  gfx9 rule (LLVM GCNHazardRecognizer, VmemSgprWaitStates): a VMEM instruction that
reads an SGPR written by a VALU instruction (v_readlane / v_readfirstlane / v_cmp ...) needs 5 wait states in between, or it may read
the OLD SGPR value.  This script walks a `hipcc -S --cuda-device-only` listing and reports every inline-asm VMEM whose scalar
operands were written by a VALU instruction fewer than 5 wait states earlier.     usage: isa_asm_hazards.py file.s [--all]"""
import re
import sys

NEED = 5
src = sys.argv[1]
lines = open(src).read().split("\n")
kern = None
hist = []            # (text, wait states it contributes)
in_asm = False
viol = {}
total = {}

def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def valu_sgpr_writes(ins):
    op, *rest = ins.replace(",", " ").split()
    if not op.startswith("v_"):
        return set()
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return sregs(rest[0])
    if op.startswith(("v_cmp", "v_cmpx")) and op.endswith("_e64"):
        return sregs(rest[0])
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(rest) > 1:
        return sregs(rest[1])
    return set()

for ln, raw in enumerate(lines, 1):
    s = raw.strip()
    m = re.match(r"^(_Z\S+):", raw)
    if m:
        kern = m.group(1); hist = []; continue
    if not s or s.startswith((".", ";")) and not s.startswith(";;#ASM"):
        continue
    if s.startswith(";;#ASMSTART"):
        in_asm = True; continue
    if s.startswith(";;#ASMEND"):
        in_asm = False; continue
    if re.match(r"^\.?LBB", s):
        hist = []          # block boundary: unknown predecessor, be quiet (the writers sit right in front of the asm in practice)
        continue
    ins = s.split(";")[0].strip()
    if not ins:
        continue
    op = ins.split()[0]
    if in_asm and op.startswith(("global_load", "buffer_load", "global_store")):
        total[kern] = total.get(kern, 0) + 1
        used = set()
        for tok in ins.replace(",", " ").split()[1:]:
            used |= sregs(tok)
        ws = 0
        for text, w in reversed(hist):
            wr = valu_sgpr_writes(text)
            if wr & used:
                if ws < NEED:
                    viol.setdefault(kern, []).append((ln, ws, text, ins))
                break
            ws += w
            if ws >= NEED:
                break
    w = 1
    if op == "s_nop":
        w = int(ins.split()[1]) + 1
    hist.append((ins, w))
    if len(hist) > 16:
        hist.pop(0)

bad = 0
for k in sorted(total):
    v = viol.get(k, [])
    bad += len(v)
    if v or "--all" in sys.argv:
        print(f"{k}: {len(v)} of {total[k]} inline-asm VMEM reads of a VALU-written SGPR within < {NEED} wait states")
        for ln, ws, text, ins in v[:6]:
            print(f"    line {ln}: {ws} wait states after `{text}` -> `{ins}`")
print(f"{bad} violations in {len(viol)} kernels of {len(total)} with inline-asm VMEM")
sys.exit(1 if bad else 0)
