#!/usr/bin/env python3
"""Cut one kernel out of a `hipcc -S --cuda-device-only` listing and summarise its hottest loop (the time loop): instruction
counts per class and every s_waitcnt / s_barrier with its line.   usage: isa_loop.py file.s <mangled-name substring> [--dump]"""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
print(f"kernel at lines {start + 1}..{end + 1}: {len(body)} lines")
# basic blocks and back edges
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            loops.append((labels[tgt], i))
if not loops:
    sys.exit("no loop found")
lo, hi = max(loops, key=lambda p: p[1] - p[0])
loop = [l.strip() for l in body[lo:hi + 1] if l.strip() and not l.strip().startswith((";", ".")) and not re.match(r"^\.LBB", l)]
print(f"largest loop: body lines {lo}..{hi} ({len(loop)} instructions)")
cls = {}
def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log"): return "valu_trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "vmem_wr"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
for l in loop:
    op = l.split()[0]
    cls[classify(op)] = cls.get(classify(op), 0) + 1
print("  ".join(f"{k}={v}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
meta = [l for l in lines[end:end + 120] if re.search(r"vgpr_count|sgpr_count|scratch|ScratchSize|NumVgprs|NumAgprs|Occupancy|LDSByteSize", l)]
print("\n".join(m.strip() for m in meta[:10]))
if "--waits" in sys.argv or "--dump" in sys.argv:
    for i, l in enumerate(body[lo:hi + 1]):
        s = l.strip()
        if "--dump" in sys.argv or s.startswith(("s_waitcnt", "s_barrier", "global_", "s_cbranch", "s_branch", ".LBB", "ds_")):
            print(f"{lo + i:6d}  {s}")
