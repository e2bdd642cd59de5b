"""K11 (csrc/psnode_linear_rows.hip) and the row MLPs built on it (round 6): the encoders / decoders of the direct_encode models at the
hidden widths K3b does not carry -- the scripts' argparse default --hidden 128 (neural_00_ODE_02_direct_encode.py:64-69, 160-162) --
forward and backward without a library GEMM, against torch in fp64."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _err(a, b):
    b = b.double().cpu()
    return float((a.double().cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("rows", [1, 33, 1000])
@pytest.mark.parametrize("K,N", [(128, 128), (8, 128), (2, 128), (128, 8), (128, 2), (16, 16), (100, 36), (20, 100), (5, 7)])
@pytest.mark.parametrize("transposed", [False, True])
def test_linear_rows_matches_fp64(rows, K, N, transposed):
    from py_psnode_amd import fused
    g = torch.Generator().manual_seed(rows + 17 * K + N)
    X = torch.randn(rows, K, generator=g).cuda()
    W = (torch.randn(K, N, generator=g) if transposed else torch.randn(N, K, generator=g)).cuda() * 0.3
    b = torch.randn(N, generator=g).cuda()
    H = (torch.randn(rows, N, generator=g).cuda() * 0.7)
    Hh = torch.nn.functional.elu(H)
    Wm = W.t() if transposed else W
    pre = X.double() @ Wm.double().t() + b.double()
    assert _err(fused.linear_rows(X, W, b, transposed=transposed), pre) <= 2e-6
    assert _err(fused.linear_rows(X, W, b, epi=1, transposed=transposed), torch.nn.functional.elu(pre)) <= 2e-6
    dgrad = torch.where(H.double() > 0, torch.ones_like(H.double()), H.double().exp())
    nob = X.double() @ Wm.double().t()
    assert _err(fused.linear_rows(X, W, None, epi=2, hh=Hh, transposed=transposed), nob * dgrad) <= 5e-6


@pytest.mark.parametrize("din,H,dout,need_gin", [(8, 128, 128, False), (2, 128, 128, False), (128, 128, 8, True), (128, 128, 2, True), (8, 100, 100, False),
                                                 (100, 100, 8, True), (3, 36, 36, True)])
def test_wide_row_mlp_autograd_matches_torch_fp64(din, H, dout, need_gin):
    """fused.mlp_rows_autograd at the widths K3b does not carry: forward and every gradient against the same nn.Sequential in fp64, on a
    time-major VIEW of a [B,T,D] batch (what the models pass) and with a non-trivial upstream gradient."""
    from py_psnode_amd import fused
    torch.manual_seed(din * 7 + H + dout)
    seq = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).cuda()
    ref = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in seq.state_dict().items()})
    B, T = 37, 23
    x_bt = (0.5 * torch.randn(B, T, din)).cuda()
    x = x_bt.permute(1, 0, 2).detach().requires_grad_(need_gin)
    xr = x_bt.permute(1, 0, 2).double().cpu().detach().requires_grad_(need_gin)
    G = torch.randn(T, B, dout).cuda()
    assert fused.rows_layers_of(seq, x, allow_grad=True) is not None
    y = fused.mlp_rows_autograd(seq, x)
    yr = ref(xr)
    assert y.shape == (T, B, dout) and _err(y, yr) <= 5e-6
    (y * G).sum().backward()
    (yr * G.double().cpu()).sum().backward()
    for (n1, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        assert _err(p.grad, q.grad) <= 2e-5, n1
    if need_gin:
        assert _err(x.grad, xr.grad) <= 2e-5
    with torch.no_grad():       # the no-grad route: K11 twice, no autograd node
        y2 = fused.mlp_rows(fused.sequential_layers(seq), x.detach())
    assert torch.equal(y2, y.detach())


def test_hidden128_model_training_step_has_no_library_gemm_in_its_row_mlps():
    """models.ODE_Model(direct_encode, hidden 128): the encoders / decoders take the K11 / K10 route (forward values and parameter
    gradients equal to the plain nn.Sequential route to rounding)."""
    from py_psnode_amd import fused, models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(0)
    B, T = 24, 12
    m = models.ODE_Model(8, 2, 128, direct_encode=True, solver=nd.Euler()).cuda()
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).cuda()
    x, z = (0.1 * torch.randn(B, T, 8)).cuda(), (0.1 * torch.randn(B, T, 2)).cuda()
    ev, zj = -torch.ones(B, 2, 1).cuda(), torch.zeros(B, 2, 2).cuda()
    calls = []
    orig = fused.rows._WideRowsMlp.apply if hasattr(fused, "rows") else None
    out = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
    (out[0].sum() + out[1].sum()).backward()
    g_fused = {n: p.grad.clone() for n, p in m.named_parameters()}
    # the same step with the row MLPs as plain modules (rows_layers_of refuses): library route
    m.zero_grad()
    keep = fused.rows_layers_of
    import py_psnode_amd.fused as F_
    import py_psnode_amd.fused.rows as R_
    try:
        F_.rows_layers_of = lambda *a, **k: None
        R_.rows_layers_of = F_.rows_layers_of
        out2 = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
        (out2[0].sum() + out2[1].sum()).backward()
    finally:
        F_.rows_layers_of = keep
        R_.rows_layers_of = keep
    assert _err(out[0], out2[0]) <= 2e-5 and _err(out[1], out2[1]) <= 2e-5
    for n, p in m.named_parameters():
        assert _err(g_fused[n], p.grad) <= 5e-4, n
