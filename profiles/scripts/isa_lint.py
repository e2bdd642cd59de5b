#!/usr/bin/env python3
"""ISA lint for the three code-generation problems round 4 root-caused in libpsnode_hip.so (DESIGN.md "Round 4: the two fenced
defects").  Input: `hipcc -S --cuda-device-only` listings (one per translation unit).  Checks, per kernel:

  A  spill-under-exec   a VGPR spill store / reload (scratch_store / scratch_load "Folded Spill / Reload") issued while EXEC may be
                        partial.  The spill saves only the ACTIVE lanes; the reload under full EXEC then hands the others garbage.
                        [defect (a): K4f <Midpoint, NZM=0, 8 waves, recompute> spilled `l & 15` inside the `4 + g < x_dim` arm of
                        load_x2 -- EXEC = 0 there at x_dim = 8 -- and the epilogue's `i < n` predicates read scratch garbage]
  B  mfma-edge          a non-MFMA instruction at a branch TARGET touches the destination of a v_mfma issued just before the branch
                        with fewer wait states in between than the MFMA's result latency.  The compiler pads the fall-through path
                        (s_nop) but missed the taken edge.  [defect (b): `if (a.gis)` of add_gis was placed between the last MFMA of
                        the AE head's third layer and the v_pk_add that sums its two accumulator chains; with grad_is = NULL the
                        branch is taken and the add reads registers 2..3 of the accumulator one instruction behind the MFMA]
  C  asm-vmem-sgpr      an inline-asm VMEM instruction reads an SGPR written by a VALU instruction < 5 wait states earlier
                        (the hazard recognizer does not look inside inline asm).  [found on the way: the LDS-DMA of K4f / K7f]

  D  asm-mfma-operand   an inline-asm MFMA reads as A / B operand a VGPR written by a VALU instruction < 2 wait states earlier (round 6:
                        K4x's gradient blocks carry a leading s_nop only where a VALU producer can precede them)

usage: isa_lint.py file.s [file.s ...] [--verbose]      exit code 1 if anything is flagged"""
import re
import sys

MFMA_BETWEEN = {"16x16x4": 10, "4x4x1": 4, "32x32x2": 18, "32x32x1": 18, "16x16x1": 10}   # wait states the compiler itself pads with
VERBOSE = "--verbose" in sys.argv
files = [a for a in sys.argv[1:] if not a.startswith("--")]


def regs(tok, kind="v"):
    tok = tok.strip().rstrip(",")
    m = re.match(rf"^{kind}\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(rf"^{kind}(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(ins):
    parts = ins.split(None, 1)
    return [p.strip() for p in parts[1].split(",")] if len(parts) > 1 else []


def ws_of(ins):
    op = ins.split()[0]
    return int(ins.split()[1]) + 1 if op == "s_nop" else 1


def valu_sgpr_writes(ins):
    op = ins.split()[0]
    ops = operands(ins)
    if not op.startswith("v_") or not ops:
        return set()
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return regs(ops[0], "s")
    if op.startswith("v_cmp") and op.endswith("_e64"):
        return regs(ops[0], "s")
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
        return regs(ops[1], "s")
    return set()


def lint_kernel(name, body, report):
    """body: list of (line number, text) of one kernel"""
    ins = []          # (ln, text, in_asm)
    labels = {}
    in_asm = False
    for ln, raw in body:
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins); continue
        if not s or s.startswith((";", ".")):
            continue
        comment = s.split(";", 1)[1] if ";" in s else ""
        t = s.split(";")[0].strip()
        if t:
            ins.append((ln, t, in_asm, comment))

    # ---- A: spills under partial EXEC.  EXEC regions are tracked over the control-flow graph (the backend lays blocks out of order: a
    # region's closing `s_or_b64 exec, exec, s[save]` may sit ABOVE its opening s_and_saveexec in the listing): the state at every
    # instruction is the stack of regions open there -- a region = the line of the instruction that narrowed EXEC (s_and_saveexec /
    # s_andn2_saveexec / exec &= .. / s_mov_b64 exec, <mask>), keyed by the SGPR pair that holds the saved mask; the else arm
    # (s_or_saveexec .. ; s_xor_b64 exec ..) replaces the region of its `then` arm; `s_or_b64 exec, exec, s[save]` closes the region saved
    # there.  A spill slot (scratch offset, or an AGPR written by v_accvgpr_write: EXEC-masked as well) is flagged when one of its stores
    # sits in a region that one of its reloads is NOT inside of: that reload runs with lanes the store never saved.
    n_ins = len(ins)
    leaders = {0} | set(labels.values())
    for i, (_, t, _, _) in enumerate(ins):
        if t.split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < n_ins)
    blk_of = {}
    for bi, st_ in enumerate(leaders):
        en = leaders[bi + 1] if bi + 1 < len(leaders) else n_ins
        for i in range(st_, en):
            blk_of[i] = bi
    state_at = [None] * n_ins        # tuple of (region id, save register)
    entry = {0: ()}
    work = [0]
    def step(state, i):
        ln, t, _, _ = ins[i]
        op = t.split()[0]
        ops = operands(t)
        st = list(state)
        if op == "s_andn2_saveexec_b64" and any(q[1] == ops[1] for q in st):
            st = [q for q in st if q[1] != ops[1]]                         # straight flip to the else arm: EXEC = else lanes (ops[1] & ~EXEC),
            st.append((ln, ops[0]))                                         # ops[0] = the then lanes, closed by `s_or_b64 exec, exec, ops[0]`
        elif op in ("s_and_saveexec_b64", "s_andn2_saveexec_b64"):
            st.append((ln, ops[0]))
        elif op == "s_xor_b64" and len(ops) == 3 and ops[1] == "exec" and ops[0] != "exec":
            st = [(q[0], ops[0]) if q[1] == ops[2] else q for q in st]        # else-mask of the region saved in ops[2]: re-keyed
        elif op == "s_or_saveexec_b64":                                    # end of the `then` arm: EXEC = the whole enclosing mask again ...
            st = [q for q in st if q[1] != ops[0]]
        elif op in ("s_xor_b64", "s_andn2_b64") and len(ops) == 3 and ops[0] == "exec" and ops[1] == "exec":
            st.append((ln, ops[2]))                                         # ... until `exec ^= reg` enters the else arm
        elif op in ("s_and_b64", "s_andn2_b64") and ops and ops[0] == "exec":
            st.append((ln, "?"))
        elif op == "s_mov_b64" and ops and ops[0] == "exec":
            if ops[1] == "-1":
                st = []
            elif any(q[1] == ops[1] for q in st):                           # restore of a saved mask
                while st and st[-1][1] != ops[1]:
                    st.pop()
                st.pop()
            else:
                st.append((ln, "?"))                                        # exec = a mask computed as exec & cond
        elif op == "s_or_b64" and ops and ops[0] == "exec":
            if any(q[1] == ops[2] for q in st):
                while st and st[-1][1] != ops[2]:
                    st.pop()
                st.pop()
            elif st and st[-1][1] == "?":
                st.pop()
        return tuple(st)
    seen = set()
    while work:
        bi = work.pop()
        if bi in seen:
            continue
        seen.add(bi)
        st_ = leaders[bi]
        en = leaders[bi + 1] if bi + 1 < len(leaders) else n_ins
        state = entry[bi]
        for i in range(st_, en):
            state_at[i] = state
            state = step(state, i)
        last = ins[en - 1][1]
        lop = last.split()[0]
        succ = []
        if lop.startswith("s_cbranch") or lop == "s_branch":
            tgt = operands(last)[0]
            if tgt in labels and labels[tgt] < n_ins:
                succ.append(blk_of[labels[tgt]])
        if not (lop == "s_branch" or lop.startswith(("s_endpgm", "s_setpc"))) and en < n_ins:
            succ.append(blk_of[en])
        for sb_ in succ:
            if sb_ not in entry:
                entry[sb_] = state
                work.append(sb_)
    slots = {}      # scratch offset / AGPR -> {"st": [(ln, regions, text)], "ld": [...]}
    for i, (ln, t, _, comment) in enumerate(ins):
        if state_at[i] is None:
            continue
        op = t.split()[0]
        regions = tuple(q[0] for q in state_at[i])
        if op.startswith(("scratch_store", "scratch_load")):
            m = re.search(r"offset:(\d+)", t)
            off = int(m.group(1)) if m else 0
            width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(op.split("_")[-1], 1)
            for q in range(width):
                rec = slots.setdefault(off + 4 * q, {"st": [], "ld": []})
                rec["st" if op.startswith("scratch_store") else "ld"].append((ln, regions, t))
        elif op == "v_accvgpr_write_b32":
            rec = slots.setdefault(operands(t)[0], {"st": [], "ld": []})
            rec["st"].append((ln, regions, t))
        elif op == "v_accvgpr_read_b32":
            src = operands(t)[1]
            if src in slots:
                slots[src]["ld"].append((ln, regions, t))
    # A reload is fine when SOME earlier store of its slot ran in a region it is inside of (every lane of the reload has been saved once;
    # narrower stores behind it -- the arms of an if / else that each define the value anew -- only refresh their own lanes).  Flagged:
    # a reload whose slot has ONLY been stored under narrower EXEC.
    flagged = set()
    for off, rec in sorted(slots.items(), key=lambda kv: str(kv[0])):
        for lnl, sld, tl in rec["ld"]:
            covering = [q for q in rec["st"] if q[0] < lnl and sld[:len(q[1])] == q[1]]
            partial = [q for q in rec["st"] if sld[:len(q[1])] != q[1]]
            if partial and not covering and (partial[0][0], lnl) not in flagged:
                lns, sst, ts = partial[0]
                flagged.add((lns, lnl))
                report("A spill-under-exec" if ts.startswith("scratch") else "A' agpr-copy-under-exec", name, lns, f"`{ts}` runs inside EXEC region(s) opened at line(s) {list(sst)} (only the lanes active there "
                                                        f"are saved) and is the only store `{tl}` at line {lnl} (regions {list(sld)}) can see")

    # ---- B: MFMA result touched at a branch target too early
    for i, (ln, t, _, _) in enumerate(ins):
        op = t.split()[0]
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        tgt = operands(t)[0] if operands(t) else None
        if tgt not in labels:
            continue
        pend = []      # (dst regs, wait states still needed at the target's first instruction, text)
        ws = 1         # the branch itself
        j = i - 1
        while j >= 0 and ws < 20:
            tj = ins[j][1]
            oj = tj.split()[0]
            if oj.startswith("v_mfma"):
                need = next((v for k, v in MFMA_BETWEEN.items() if k in oj), 10)
                d = {("v", q) for q in regs(operands(tj)[0])} | {("a", q) for q in regs(operands(tj)[0], "a")}
                if ws < need and not any(d & q[0] for q in pend):      # (an older MFMA into the same registers is ordered by the newer one)
                    pend.append((d, need - ws, tj))
            if oj.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")) and j != i - 0:
                pass
            ws += ws_of(tj)
            j -= 1
        if not pend:
            continue
        k = labels[tgt]
        ws = 0
        while k < len(ins) and ws < 20 and pend:
            lk, tk, _, _ = ins[k]
            ok = tk.split()[0]
            touched = set()
            ops = operands(tk)
            both = lambda o: {("v", q) for q in regs(o)} | {("a", q) for q in regs(o, "a")}
            if ok.startswith("v_mfma"):
                for o in ops[1:3]:            # A / B operands (the accumulator operand may follow back to back)
                    touched |= both(o)
            elif ok.startswith(("v_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
                for o in ops:
                    touched |= both(o.split()[0] if o else o)
            for d, left, tm in list(pend):
                if touched & d and ws < left:
                    report("B mfma-edge", name, lk, f"`{tk}` at {tgt} is {ws + (MFMA_BETWEEN.get('16x16x4') - left)} wait states behind `{tm}` "
                                                    f"on the taken edge of `{t}` (line {ln}); {left - ws} more needed")
                    pend.remove((d, left, tm))
            if ok.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            ws += ws_of(tk)
            k += 1

    # ---- C: inline-asm VMEM reading a VALU-written SGPR
    hist = []
    for ln, t, asm, _ in ins:
        op = t.split()[0]
        if asm and op.startswith(("global_load", "buffer_load", "global_store")):
            used = set()
            for o in operands(t):
                used |= regs(o.split()[0], "s")
            ws = 0
            for tx, w in reversed(hist):
                if valu_sgpr_writes(tx) & used:
                    if ws < 5:
                        report("C asm-vmem-sgpr", name, ln, f"`{t}` {ws} wait states after `{tx}` (needs 5)")
                    break
                ws += w
                if ws >= 5:
                    break
        hist.append((t, ws_of(t)))
        if len(hist) > 16:
            hist.pop(0)

    # ---- D: an inline-asm MFMA reads, as its A / B operand, a VGPR that a (non-MFMA) VALU instruction wrote < 2 wait states earlier.  The
    # hazard recognizer pads compiler-visible MFMAs; it does not look inside inline asm, so an asm block that relies on "the previous
    # instruction is my own MFMA" (K4x's gradient blocks without a leading s_nop, round 6) must be checked in the code that ships: a
    # spilled operand coming back through v_accvgpr_read right in front of the block would be exactly that.
    hist = []
    for ln, t, asm, _ in ins:
        op = t.split()[0]
        if asm and op.startswith("v_mfma"):
            ops = operands(t)
            used = set()
            for o in ops[1:3]:
                used |= regs(o.split()[0])
            ws = 0
            for tx, w, was_asm in reversed(hist):
                ox = tx.split()[0]
                if ox.startswith("v_") and not ox.startswith("v_mfma"):
                    dst = operands(tx)
                    if dst and regs(dst[0].split()[0]) & used and ws < 2:
                        report("D asm-mfma-operand", name, ln, f"`{t}` {ws} wait states after `{tx}` (needs 2)")
                        break
                ws += w
                if ws >= 2:
                    break
        hist.append((t, ws_of(t), asm))
        if len(hist) > 8:
            hist.pop(0)


total = {}
kernels = 0


def report(kind, name, ln, msg):
    total.setdefault(kind, []).append((name, ln, msg))


for path in files:
    lines = open(path).read().split("\n")
    cur, body = None, []
    for ln, raw in enumerate(lines, 1):
        m = re.match(r"^(_Z\S+):", raw)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append((ln, raw))
            if raw.strip().startswith("s_endpgm"):
                # kernels end with s_endpgm; helper code after it (none here) is ignored
                pass
            if raw.strip().startswith(".end_amdhsa_kernel") or raw.strip().startswith(".Lfunc_end"):
                kernels += 1
                lint_kernel(f"{path.split('/')[-1]}:{cur}", body, report)
                cur = None

rc = 0
GATING = ("A spill-under-exec", "B mfma-edge", "C asm-vmem-sgpr", "A' agpr-copy-under-exec", "D asm-mfma-operand")
# (A': VGPR -> AGPR copies are EXEC-masked like scratch stores -- a value parked in an AGPR from inside a divergent arm and read back
#  outside of it is the same defect.  Gating since the region model follows every structure the backend emits for these kernels; its
#  first real catch was a row index of K7f's round-4 code, parked in a25 inside one arm of a nested ?:.)
for kind in GATING:
    items = total.get(kind, [])
    print(f"[{kind}] {len(items)} site(s) in {len({n for n, _, _ in items})} kernel(s)")
    shown = {}
    for name, ln, msg in items:
        shown[name] = shown.get(name, 0) + 1
        if VERBOSE or shown[name] <= 2:
            print(f"    {name.split(':')[0]}:{ln}  {name.split(':', 1)[1][:110]}\n        {msg}")
    rc |= bool(items) and kind in GATING
print(f"{kernels} kernels in {len(files)} file(s) checked")
sys.exit(rc)
