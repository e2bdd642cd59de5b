"""K10 (csrc/psnode_gemm_tn.hip, round 6): the tall-skinny contraction over rows C = A^T B (+ column sums of A) on MFMA, which replaces
the library GEMMs of the latent-wide backward (fused/latent.py; what loss.backward() forms for de_func / ae_func of the direct_encode
models at hidden widths other than 16 / 64, neural_00_ODE_02_direct_encode.py:160-162, 267-275) -- against torch in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", [0, 1, 3, 5, 257, 1000, 100003])
@pytest.mark.parametrize("M,N", [(128, 128), (64, 64), (4, 128), (128, 8), (100, 36), (12, 12), (68, 64), (64, 68)])
def test_gemm_tn_matches_fp64(rows, M, N):
    from py_psnode_amd import fused
    from py_psnode_amd.fused._common import gemm_tn
    g = torch.Generator().manual_seed(rows * 131 + M * 7 + N)
    A, B = torch.randn(rows, M, generator=g).cuda(), torch.randn(rows, N, generator=g).cuda()
    out = gemm_tn(A, B, want_colsum=True)
    assert out is not None
    C, cs = out
    ref = (A.double().t() @ B.double()).cpu()
    refs = A.double().sum(0).cpu()
    tol = 2e-6 * max(1.0, float(rows) ** 0.5)
    assert C.shape == (M, N) and float((C.double().cpu() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert float((cs.double().cpu() - refs).abs().max()) <= tol * max(1.0, float(refs.abs().max()))
    C2 = gemm_tn(A, B)
    assert torch.equal(C2, C), "deterministic: per-workgroup partials summed in a fixed order"


def test_gemm_tn_row_strides_and_refusals():
    from py_psnode_amd.fused._common import gemm_tn
    g = torch.Generator().manual_seed(5)
    big = torch.randn(5000, 256, generator=g).cuda()
    A, B = big[:, 0:64], big[:, 128:228]            # row-strided views: lda = ldb = 256, 16-byte aligned column offsets
    C = gemm_tn(A, B)
    ref = (A.double().t() @ B.double()).cpu()
    assert float((C.double().cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    assert gemm_tn(big[:, 0:6], big[:, 8:16]) is None            # M % 4 != 0: outside the class, the caller falls back
    assert gemm_tn(big[:, 0:132], big[:, 0:8]) is None           # M > 128
    assert gemm_tn(big[:, 1:65], big[:, 0:8]) is None            # rows not 16-byte aligned
    assert gemm_tn(big.double()[:, 0:8], big.double()[:, 0:8]) is None


def test_gemm_tn_full_size_rows_of_the_hidden128_latent_backward():
    """The contraction K9w needs at the scripts' argparse default: 4096 x 1000 Euler rows of width 128 (134 GFLOP), against chunked fp64."""
    from py_psnode_amd.fused._common import gemm_tn
    R, H = 4096 * 1000, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    A, B = torch.randn(R, H, device="cuda", generator=g) * 0.1, torch.randn(R, H, device="cuda", generator=g) * 0.1
    C, cs = gemm_tn(A, B, want_colsum=True)
    ref = torch.zeros(H, H, dtype=torch.float64, device="cuda")
    for c0 in range(0, R, 1 << 19):
        ref += A[c0:c0 + (1 << 19)].double().t() @ B[c0:c0 + (1 << 19)].double()
    assert float((C.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-3
    assert float((cs.double() - A.double().sum(0)).abs().max()) <= 1e-3
