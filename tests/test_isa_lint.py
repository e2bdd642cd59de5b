"""profiles/scripts/isa_lint.py -- the build-time check for the three code-generation patterns behind round 3's wrong gradients (DESIGN.md
"Round 4: the two fenced defects").  (1) the lint recognises each pattern on a minimal listing cut from the real failing kernels;
(2) the listings the shipped libpsnode_hip.so was assembled from are clean (`make` keeps them under build/obj/ and fails on a finding;
this test re-runs the lint on them when they are present -- they do not travel to the GPU box)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "profiles", "scripts", "isa_lint.py")

HEAD = "_ZN6psnode4testEv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"

# defect (a), K4f <Midpoint, NZM = 0, 8 waves, recompute>: the only store of the slot sits in the `then` arm of a divergent if
SPILL_UNDER_EXEC = HEAD + """\tv_and_b32_e32 v96, 15, v0
\ts_and_saveexec_b64 s[26:27], s[30:31]
\ts_xor_b64 s[30:31], exec, s[26:27]
\tv_add_u32_e32 v191, 16, v190
\tscratch_store_dword off, v96, off offset:8 ; 4-byte Folded Spill
\ts_or_saveexec_b64 s[30:31], s[30:31]
\tv_mov_b32_e32 v70, 0
\ts_xor_b64 exec, exec, s[30:31]
\ts_cbranch_execz .LBB0_36
\tglobal_load_dword v71, v2, s[98:99]
.LBB0_36:
\ts_or_b64 exec, exec, s[30:31]
\tv_add_f32_e64 v96, v158, v96
\tscratch_load_dword v96, off, off offset:8 ; 4-byte Folded Reload
\ts_waitcnt vmcnt(0)
\tv_cmp_gt_i32_e32 vcc, s56, v96
""" + TAIL
# the legitimate shape: every arm of an if / else defines the value anew and stores it, behind a full-EXEC store
SPILL_PER_ARM = HEAD + """\tscratch_store_dwordx2 off, v[34:35], off ; 8-byte Folded Spill
\ts_and_saveexec_b64 s[6:7], vcc
\ts_xor_b64 s[6:7], exec, s[6:7]
\tv_subrev_u32_e32 v34, s46, v33
\tscratch_store_dwordx2 off, v[34:35], off ; 8-byte Folded Spill
\ts_andn2_saveexec_b64 s[6:7], s[6:7]
\tv_subrev_u32_e32 v34, s22, v33
\tscratch_store_dwordx2 off, v[34:35], off ; 8-byte Folded Spill
\ts_or_b64 exec, exec, s[6:7]
\tscratch_load_dwordx2 v[26:27], off, off ; 8-byte Folded Reload
""" + TAIL
# defect (b), K7w: `if (grad_is)` placed between the last MFMA of a layer and the add that sums the two accumulator chains
MFMA_EDGE = HEAD + """\tv_mfma_f32_16x16x4_f32 v[0:3], v83, v36, v[0:3]
\tv_mfma_f32_16x16x4_f32 v[14:17], v73, v37, v[14:17]
\ts_cbranch_scc1 .LBB0_147
\tv_cmp_eq_u32_e32 vcc, 2, v131
\ts_nop 9
.LBB0_147:
\tv_pk_add_f32 v[16:17], v[2:3], v[16:17]
""" + TAIL
MFMA_EDGE_PADDED = MFMA_EDGE.replace(".LBB0_147:\n", ".LBB0_147:\n\ts_nop 9\n")
# the LDS-DMA of K4f / K7f as rounds 2-3 had it: 4 wait states between the v_readlane of the base's high half and the load
ASM_VMEM = HEAD + """\tv_readlane_b32 s6, v255, 11
\tv_readlane_b32 s7, v255, 12
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\ts_mov_b32 s2, m0
\ts_mov_b32 m0, s75
\ts_nop 0
\tglobal_load_lds_dwordx4 v183, s[6:7]
\ts_mov_b32 m0, s2
\t;;#ASMEND
""" + TAIL
ASM_VMEM_FIXED = ASM_VMEM.replace("s_nop 0", "s_nop 2")

# round 6, K4x: a gradient block without a leading s_nop right behind the v_accvgpr_read that brings a spilled operand back
ASM_MFMA_OPERAND = HEAD + """\tv_accvgpr_read_b32 v5, a173
\t;;#ASMSTART
\tv_mfma_f32_4x4x1_16b_f32 a[0:3], v5, v36, a[0:3] cbsz:4 abid:2
\t;;#ASMEND
""" + TAIL
ASM_MFMA_OPERAND_FIXED = ASM_MFMA_OPERAND.replace("\t;;#ASMSTART\n", "\t;;#ASMSTART\n\ts_nop 1\n")


def _lint(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    r = subprocess.run([sys.executable, LINT, str(f)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_lint_flags_a_spill_that_only_saves_the_lanes_of_a_divergent_arm(tmp_path):
    rc, out = _lint(tmp_path, SPILL_UNDER_EXEC)
    assert rc == 1 and "[A spill-under-exec] 1 site" in out, out
    rc, out = _lint(tmp_path, SPILL_PER_ARM)
    assert rc == 0 and "[A spill-under-exec] 0 site" in out, out


def test_lint_flags_an_mfma_result_read_across_a_taken_branch_edge(tmp_path):
    rc, out = _lint(tmp_path, MFMA_EDGE)
    assert rc == 1 and "[B mfma-edge] 2 site" in out, out     # both accumulator chains are read too early
    rc, out = _lint(tmp_path, MFMA_EDGE_PADDED)
    assert rc == 0 and "[B mfma-edge] 0 site" in out, out


def test_lint_flags_inline_asm_vmem_behind_a_valu_written_sgpr(tmp_path):
    rc, out = _lint(tmp_path, ASM_VMEM)
    assert rc == 1 and "[C asm-vmem-sgpr] 1 site" in out, out
    rc, out = _lint(tmp_path, ASM_VMEM_FIXED)
    assert rc == 0, out


def test_lint_flags_an_inline_asm_mfma_behind_the_valu_write_of_its_operand(tmp_path):
    rc, out = _lint(tmp_path, ASM_MFMA_OPERAND)
    assert rc == 1 and "[D asm-mfma-operand] 1 site" in out, out
    rc, out = _lint(tmp_path, ASM_MFMA_OPERAND_FIXED)
    assert rc == 0 and "[D asm-mfma-operand] 0 site" in out, out


def test_shipped_listings_are_clean():
    lists = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if len(lists) < 20:
        import pytest
        pytest.skip("no device listings here (build/obj/ is made by `make` in the build container and does not travel)")
    r = subprocess.run([sys.executable, LINT] + lists, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    for kind in ("[A spill-under-exec] 0 site", "[B mfma-edge] 0 site", "[C asm-vmem-sgpr] 0 site", "[D asm-mfma-operand] 0 site"):
        assert kind in r.stdout, r.stdout[-2000:]
