"""GPU parity of the row-MLP backward kernel (K3b backward, psnode_mlp_rows_backward_f32) against fp64 autograd of the same
nn.Sequential(Linear, ELU, Linear): grad of the input rows and of all four parameter tensors, every (in, hidden, out)
class the direct_encode models use (encoders in<=16 -> H -> H, decoders H -> H -> out<=16, H in {16, 64}), ragged row
counts, strided 3-D inputs; and the model-level route (ODE_02 / DAE_02 training steps take it).

Tolerance: parameter gradients are fp32 sums over up to 4100 rows with a different summation order: 2e-5 of each
tensor's max; input gradients (no reduction over rows) 2e-6."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= rtol * max(scale, 1e-12), f"{what}: err {err:.3e} vs scale {scale:.3e}"


SHAPES = [(8, 16, 16), (2, 16, 16), (16, 16, 8), (16, 16, 2), (3, 16, 5), (1, 16, 16), (6, 16, 16),
          (8, 64, 64), (2, 64, 64), (64, 64, 8), (64, 64, 2), (5, 64, 64)]


@pytest.mark.parametrize("rows", [1, 17, 4100])
@pytest.mark.parametrize("din,H,dout", SHAPES)
def test_rows_backward_matches_fp64_autograd(din, H, dout, rows):
    from py_psnode_amd import fused
    torch.manual_seed(din * 100 + dout)
    seq = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout))
    inp, gout = torch.randn(rows, din), torch.randn(rows, dout)
    seq64 = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).double()
    seq64.load_state_dict({k: v.double() for k, v in seq.state_dict().items()})
    x64 = inp.double().requires_grad_(True)
    (seq64(x64) * gout.double()).sum().backward()
    layers = [(seq[0].weight.detach().cuda(), seq[0].bias.detach().cuda()), (seq[2].weight.detach().cuda(), seq[2].bias.detach().cuda())]
    gin, gp = fused.mlp_rows_backward(layers, inp.cuda(), gout.cuda())
    _close(gin, x64.grad, 2e-6, "grad in")
    for k, (a, p) in enumerate(zip(gp, [seq64[0].weight, seq64[0].bias, seq64[2].weight, seq64[2].bias])):
        _close(a, p.grad, 2e-5, f"grad param {k}")
    gin2, gp2 = fused.mlp_rows_backward(layers, inp.cuda(), gout.cuda(), need_grad_in=False)
    assert gin2 is None and all(torch.equal(a, b) for a, b in zip(gp, gp2)), "deterministic, grad_in optional"


def test_rows_autograd_function_on_strided_3d_input():
    """The route models._rows takes under autograd: [B,T,D] input that is a permuted view, gradient through input and parameters."""
    from py_psnode_amd import fused
    torch.manual_seed(5)
    seq = nn.Sequential(nn.Linear(8, 16), nn.ELU(), nn.Linear(16, 16)).cuda()
    ref = nn.Sequential(nn.Linear(8, 16), nn.ELU(), nn.Linear(16, 16)).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in seq.state_dict().items()})
    base = torch.randn(37, 11, 8)
    G = torch.randn(11, 37, 16)
    xg = base.cuda().requires_grad_(True)
    out = fused.mlp_rows_autograd(seq, xg.permute(1, 0, 2))
    assert type(out.grad_fn).__name__.startswith("_RowsMlp")
    (out * G.cuda()).sum().backward()
    x64 = base.double().requires_grad_(True)
    (ref(x64.permute(1, 0, 2)) * G.double()).sum().backward()
    _close(xg.grad, x64.grad, 2e-6, "grad in")
    for (n, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 2e-5, n)


@pytest.mark.parametrize("din,H,dout", [(8, 16, 16), (2, 16, 16), (3, 16, 5), (1, 16, 16), (8, 64, 64), (16, 16, 8)])
@pytest.mark.parametrize("B,T", [(1, 1), (5, 1), (1, 7), (37, 11), (130, 33)])
def test_rows_read_a_batch_major_tensor_as_time_major_rows_in_place(din, H, dout, B, T):
    """ABI 9: the [B,T,D] batch goes in as its time-major view (x.permute(1, 0, 2), neural_00_ODE_02_direct_encode.py:76) and is read where it
    lies -- two-level row addressing -- with the results of the contiguous copy (rows bit for bit; parameter gradients to rounding: another fixed tile order); no copy is made."""
    from py_psnode_amd import fused
    from py_psnode_amd.fused import rows as R
    torch.manual_seed(B * 100 + T + din)
    seq = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).cuda()
    layers = [(seq[0].weight.detach(), seq[0].bias.detach()), (seq[2].weight.detach(), seq[2].bias.detach())]
    base = torch.randn(B, T, din, device="cuda")
    view = base.permute(1, 0, 2)
    kept, rows, rstride, inner, outer = R._row_addressing(view)
    assert kept.data_ptr() == base.data_ptr() and rows == B * T
    if B > 1 and T > 1:
        assert (rstride, inner, outer) == (T * din, B, din)
    out_v, out_c = fused.mlp_rows(layers, view), fused.mlp_rows(layers, view.contiguous())
    assert out_v.shape == (T, B, dout) and torch.equal(out_v, out_c)
    G = torch.randn(T, B, dout, device="cuda")
    gin_v, gp_v = fused.mlp_rows_backward(layers, view, G)
    gin_c, gp_c = fused.mlp_rows_backward(layers, view.contiguous(), G)
    assert torch.equal(gin_v, gin_c)
    for k, (a_, b_) in enumerate(zip(gp_v, gp_c)):      # the in-place read walks 4 x 4 (grid point, trajectory) tiles, the copy 16 rows in a line:
        _close(a_, b_, 2e-6, f"param {k}: same rows, another fixed summation order")      # the same sum in another (fixed) order
    sl = base[:, :, :din][:, ::2] if T > 1 else base       # a strided slice along T: still two-level (inner stride 2 * din)
    assert torch.equal(fused.mlp_rows(layers, sl.permute(1, 0, 2)), fused.mlp_rows(layers, sl.permute(1, 0, 2).contiguous()))


@pytest.mark.parametrize("din,H,dout", [(8, 16, 16), (2, 16, 16), (16, 16, 8), (8, 64, 64), (64, 64, 2)])
def test_one_module_over_several_row_sets_is_one_autograd_node(din, H, dout):
    """models._rows_multi: x_encoder over the grid rows AND the first row (z_encoder: and the jump rows; the decoder: solution and
    reconstruction) as ONE autograd node -- every set's backward leaves its partials in one buffer, psnode_mlp_rows_reduce_f32 sums them in a
    fixed order.  Equals the per-set route (sum of the per-set parameter gradients, the same input gradients), skips an unused output, is
    deterministic, and matches fp64 autograd."""
    from py_psnode_amd import fused
    torch.manual_seed(din + dout)
    seq = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).cuda()
    B, T = 37, 11
    big = torch.randn(B, T, din, device="cuda")
    a = big.permute(1, 0, 2).requires_grad_(True) if din == 64 else big.permute(1, 0, 2)     # a strided time-major view, a [B, din] slice, a [B, 2, din] tensor
    b, c = big[:, 0], torch.randn(B, 2, din, device="cuda", requires_grad=True)
    Ga, Gb, Gc = torch.randn(T, B, dout, device="cuda"), torch.randn(B, dout, device="cuda"), torch.randn(B, 2, dout, device="cuda")

    def run(multi, use_b=True):
        seq.zero_grad(set_to_none=True)
        if c.grad is not None:
            c.grad = None
        outs = fused.mlp_rows_autograd_multi(seq, a, b, c) if multi else tuple(fused.mlp_rows_autograd(seq, q) for q in (a, b, c))
        loss = (outs[0] * Ga).sum() + (outs[2] * Gc).sum() + ((outs[1] * Gb).sum() if use_b else 0.0)
        loss.backward()
        return [o.detach().clone() for o in outs], [p.grad.clone() for p in seq.parameters()], c.grad.clone()

    o_m, g_m, gc_m = run(True)
    o_s, g_s, gc_s = run(False)
    assert all(torch.equal(x_, y_) for x_, y_ in zip(o_m, o_s)) and torch.equal(gc_m, gc_s)
    for k, (x_, y_) in enumerate(zip(g_m, g_s)):
        _close(x_, y_, 2e-6, f"param {k}: one reduction vs per-set reductions + add")
    o_m2, g_m2, _ = run(True)
    assert all(torch.equal(x_, y_) for x_, y_ in zip(g_m, g_m2)), "deterministic"
    _, g_nb, _ = run(True, use_b=False)          # an unused output: its set is skipped (no zeros are materialised)
    _, g_nb_s, _ = run(False, use_b=False)
    for k, (x_, y_) in enumerate(zip(g_nb, g_nb_s)):
        _close(x_, y_, 2e-6, f"param {k} with an unused output")
    ref = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout)).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in seq.state_dict().items()})
    ((ref(a.detach().double().cpu()) * Ga.double().cpu()).sum() + (ref(b.double().cpu()) * Gb.double().cpu()).sum()
     + (ref(c.detach().double().cpu()) * Gc.double().cpu()).sum()).backward()
    for (n, p), q in zip(seq.named_parameters(), g_m):
        _close(q, dict(ref.named_parameters())[n].grad, 2e-5, n)


@pytest.mark.parametrize("din,dout", [(8, 8), (2, 2), (5, 3), (16, 16), (1, 1), (12, 7)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (37, 11), (130, 33), (16, 16)])
def test_reconstruction_branch_in_one_kernel_each_way(din, dout, B, T):
    """K3r (psnode_recon_rows_f32 / _backward_f32): x_decoder(x_encoder(x)) of neural_00_ODE_02_direct_encode.py:87 at hidden 16 with the encoded
    rows never in memory -- forward against the two row kernels it replaces and against fp64, the parameter gradients of BOTH modules against
    fp64 autograd and against the unfused route, on the time-major view of a [B,T,D] batch (read in place) and on a contiguous copy; ragged
    row counts; deterministic."""
    from py_psnode_amd import fused
    torch.manual_seed(din * 31 + dout + B)
    enc = nn.Sequential(nn.Linear(din, 16), nn.ELU(), nn.Linear(16, 16)).cuda()
    dec = nn.Sequential(nn.Linear(16, 16), nn.ELU(), nn.Linear(16, dout)).cuda()
    base = torch.randn(B, T, din, device="cuda")
    view = base.permute(1, 0, 2)
    G = torch.randn(T, B, dout, device="cuda")
    le, ld = fused.sequential_layers(enc), fused.sequential_layers(dec)
    assert fused.recon_rows_supported(le, ld, view)

    def grads(fn, inp):
        enc.zero_grad(set_to_none=True); dec.zero_grad(set_to_none=True)
        out = fn(inp)
        (out * G).sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in (*enc.parameters(), *dec.parameters())]

    o_f, g_f = grads(lambda a: fused.recon_rows_autograd(enc, dec, a), view)
    o_c, g_c = grads(lambda a: fused.recon_rows_autograd(enc, dec, a), view.contiguous())
    o_u, g_u = grads(lambda a: fused.mlp_rows_autograd(dec, fused.mlp_rows_autograd(enc, a)), view)
    assert o_f.shape == (T, B, dout) and torch.equal(o_f, o_c) and all(torch.equal(a, b) for a, b in zip(g_f, g_c))
    assert torch.equal(o_f, fused.recon_rows(le, ld, view))
    _close(o_f, o_u, 2e-6, "x_re vs the two row kernels")
    for k, (a, b) in enumerate(zip(g_f, g_u)):
        _close(a, b, 2e-5, f"param {k} vs the unfused route")
    e64 = nn.Sequential(nn.Linear(din, 16), nn.ELU(), nn.Linear(16, 16)).double()
    d64 = nn.Sequential(nn.Linear(16, 16), nn.ELU(), nn.Linear(16, dout)).double()
    e64.load_state_dict({k: v.double().cpu() for k, v in enc.state_dict().items()})
    d64.load_state_dict({k: v.double().cpu() for k, v in dec.state_dict().items()})
    y64 = d64(e64(view.double().cpu()))
    (y64 * G.double().cpu()).sum().backward()
    _close(o_f, y64, 2e-6, "x_re vs fp64")
    for k, (a, p) in enumerate(zip(g_f, (*e64.parameters(), *d64.parameters()))):
        _close(a, p.grad, 2e-5, f"param {k} vs fp64 autograd")
    _, g_2 = grads(lambda a: fused.recon_rows_autograd(enc, dec, a), view)
    assert all(torch.equal(a, b) for a, b in zip(g_f, g_2)), "deterministic"


@pytest.mark.parametrize("events", [False, True])
@pytest.mark.parametrize("method", ["euler", "rk4"])
@pytest.mark.parametrize("tag,H,zd", [("ode02", 16, 2), ("dae02", 16, 2), ("dae02", 16, 0), ("ode02", 64, 2), ("dae02", 64, 2), ("dae02", 64, 0)])
def test_direct_encode_training_step_uses_row_kernels_and_matches_fp64(tag, H, zd, method, events):
    _direct_encode_case(tag, H, zd, method, events, 19, 9)


@pytest.mark.parametrize("B,T", [(3, 2), (17, 1), (1, 3), (33, 4)])
@pytest.mark.parametrize("tag", ["ode02", "dae02"])
def test_latent64_backward_edge_sizes(tag, B, T):
    """K9 at the edges: single step, no step at all (T = 1), single trajectory, ragged second tile."""
    _direct_encode_case(tag, 64, 2, "rk4", False, B, T)


def _direct_encode_case(tag, H, zd, method, events, B, T):
    """ODE_02 / DAE_02 models (hidden 16 and the shipped hidden 64, with and without z): loss.backward() through encoders ->
    fused latent integrator (forward K3a/K3c, backward K8 / K5 / K9) -> decoders vs the fp64 autograd walk on the CPU; the
    solver must have taken the fused route and the encoders/decoders the row kernels."""
    from py_psnode_amd import models, neural_dae as nd
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1)
    x, z, v, i = r(B, T, 8), r(B, T, zd), r(B, T, 2), r(B, T, 2)
    ev = t[:, [2, 6], :].contiguous() if (events and T > 7) else -torch.ones(B, 2, 1)
    zj, vj = r(B, 2, zd), r(B, 2, 2)
    cls = {"euler": nd.Euler, "rk4": nd.RK4}[method]
    if tag == "ode02":
        mk = lambda: models.ODE_Model(8, zd, H, direct_encode=True, solver=cls())
    else:
        mk = lambda: models.DAE_Model(8, zd, 2, 2, H, direct_encode=True, solver=cls())
    m32, m64 = mk(), mk().double()
    m64.load_state_dict({k: v_.double() for k, v_ in m32.state_dict().items()})
    m64.solver.fused = "off"
    m32 = m32.cuda()
    m32.solver.fused = "require"

    def run(model, cast, dev):
        c = lambda a: cast(a).to(dev)
        if tag == "ode02":
            outs = model(t=c(t), x=c(x), z=c(z), event_t=c(ev), z_jump=c(zj))
        else:
            outs = model(t=c(t), x=c(x), z=c(z), v=c(v), i=c(i), event_t=c(ev), z_jump=c(zj), v_jump=c(vj))
        loss = sum(((o - 0.05) ** 2).sum() for o in outs)
        loss.backward()
        return [o.detach() for o in outs], outs[-1].grad_fn

    ref, _ = run(m64, lambda a: a.double(), "cpu")
    out, gfn = run(m32, lambda a: a, "cuda")
    # the reconstruction comes off the row kernel (time-major on the HIP route: a permuted view of its output)
    node = gfn.next_functions[0][0] if type(gfn).__name__ == "PermuteBackward0" else gfn
    assert type(node).__name__.startswith(("_RowsMlp", "_ReconRows")), type(node).__name__      # (ODE_02 at hidden 16: the fused reconstruction, K3r)
    for a, b in zip(out, ref):
        _close(a, b, 1e-5, "model output")
    for (n, p), (_, q) in zip(m32.named_parameters(), m64.named_parameters()):
        if q.grad is None or p.grad is None:       # parameter unused by this graph (T = 1: no step): both None or zero
            assert (p.grad is None or float(p.grad.abs().max()) == 0.0) and (q.grad is None or float(q.grad.abs().max()) == 0.0), n
            continue
        _close(p.grad, q.grad, 5e-4, n)
