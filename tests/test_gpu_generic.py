"""The generic integrator K0 (csrc/psnode_generic.hip) in each of its three forms, against the CPU oracle: the shapes outside the
specialised integrators' classes -- x_dim > 16, z + v + i > 8, depth != 3 hidden layers, mixed and very wide layers; all of them data- or
user-defined upstream (neural_00_ODE_01_no_encode.py:293).
  register form : layers of <= 64 units, first contraction <= 128 columns (the wave's MFMA A operands stay in VGPRs);
  wide register form (ODE): 2..4 layers, hidden layers of <= 128 units = two tiles per wave on shared activation reads;
  LDS form      : every layer's weight image fits the LDS left over;
  streamed form : the layers that do not fit are read from the L2-resident workspace image one chunk ahead.
Events (with the i0 recompute), the four teacher-forcing combinations, per-trajectory clocks, ragged last tiles, T = 1 and 2."""
import pytest
import torch
import torch.nn as nn

from helpers import TOL_GPU, traj_rel_err as rel_err
from oracle import psnode_oracle as O

pytestmark = pytest.mark.gpu
METHODS = ("euler", "midpoint", "rk4")


def fused():
    from py_psnode_amd import fused as f
    return f


def dl(ls):
    return [(w.cuda(), b.cuda()) for w, b in ls]


def mk(dims, seed):
    torch.manual_seed(seed)
    return [(l.weight.detach(), l.bias.detach()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]


def ode_case(B, Tn, xd, zd, hidden, seed):
    g = torch.Generator().manual_seed(seed)
    n = xd + zd
    ls = mk([3 * n] + list(hidden) + [xd], seed)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    t = t * (0.5 + torch.rand(1, B, 1, generator=g))
    t[:, 0] = torch.arange(Tn, dtype=torch.float32).view(Tn, 1) * 0.01      # trajectory 0 = the event clock
    x = 0.1 * torch.randn(Tn, B, xd, generator=g)
    z = 0.1 * torch.randn(Tn, B, zd, generator=g)
    a0 = torch.cat((x[0], z[0]), -1)
    ev = zj = None
    if Tn > 6:
        ev = torch.stack([t[2, :, :], t[Tn - 3, :, :]], dim=1).contiguous()
        zj = 0.1 * torch.randn(B, 2, zd, generator=g)
    return ls, t, x, z, a0, ev, zj


def check_ode(B, Tn, xd, zd, hidden, method, seed=5, kernel="auto"):
    ls, t, x, z, a0, ev, zj = ode_case(B, Tn, xd, zd, hidden, seed)
    c = lambda v: None if v is None else v.cuda()
    for tx in (False, True):
        ref = O.integrate_ode(method, ls, t, x, z, a0, ev, zj, input_true_x=tx)
        out = fused().ode_integrate(method, dl(ls), c(t), c(x), c(z), c(a0), event_t=c(ev), z_jump=c(zj), input_true_x=tx, kernel=kernel)
        assert out.shape == ref.shape
        assert rel_err(out.cpu(), ref) <= TOL_GPU, (xd, zd, hidden, method, tx)


def dae_case(B, Tn, xd, zd, vd, idim, hde, hae, seed):
    g = torch.Generator().manual_seed(seed)
    n = xd + zd + vd + idim
    de, ae = mk([3 * n] + list(hde) + [xd], seed), mk([n + xd + zd + vd] + list(hae) + [idim], seed + 1)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    x, z, v, i, xi = r(Tn, B, xd), r(Tn, B, zd), r(Tn, B, vd), r(Tn, B, idim), r(B, xd)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
    ev = torch.stack([t[3, :, :], t[Tn - 2, :, :]], dim=1).contiguous() if Tn > 5 else None
    return de, ae, t, x, z, v, i, xi, a0, ev, r(B, 2, zd), r(B, 2, vd)


def check_dae(B, Tn, xd, zd, vd, idim, hde, hae, method, seed=9, combos=((False, False), (True, False), (False, True), (True, True))):
    de, ae, t, x, z, v, i, xi, a0, ev, zj, vj = dae_case(B, Tn, xd, zd, vd, idim, hde, hae, seed)
    c = lambda a: None if a is None else a.cuda()
    for tx, ti in combos:
        ref_x, ref_i = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, ev, zj if ev is not None else None, vj if ev is not None else None,
                                       input_true_x=tx, input_true_i=ti)
        xs, is_ = fused().dae_integrate(method, dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), event_t=c(ev),
                                        z_jump=c(zj) if ev is not None else None, v_jump=c(vj) if ev is not None else None, input_true_x=tx,
                                        input_true_i=ti)
        assert rel_err(xs.cpu(), ref_x) <= TOL_GPU, (tx, ti, "x")
        assert rel_err(is_.cpu(), ref_i) <= TOL_GPU, (tx, ti, "i")


# ---- register form
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,hidden", [(20, 2, (64, 64, 64)),        # x_dim > 16: first contraction 66 columns = 5 quads (QM 8)
                                          (32, 4, (64, 64, 64)),        # 108 columns = 7 quads
                                          (8, 2, (64, 64)),             # two hidden layers (QM 4)
                                          (8, 2, (48,)),                # one hidden layer, zero-padded width
                                          (17, 0, (33, 17, 64)),        # z_dim = 0, odd widths, tiles the waves do not all own
                                          (5, 3, (16, 64, 16))])
def test_register_form_ode(xd, zd, hidden, method):
    check_ode(37, 14, xd, zd, hidden, method)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("dims", [(8, 4, 6, 6), (8, 2, 1, 1), (12, 3, 4, 9), (3, 0, 5, 2)])
def test_register_form_dae(dims, method):
    """z + v + i > 8 / x_dim > 8: outside K2's classes.  Events with the i0 recompute, all four teacher-forcing combinations."""
    check_dae(21, 11, *dims, (64, 64, 64), (64, 64, 64), method)


def test_register_form_dae_differing_depths():
    check_dae(19, 9, 8, 2, 2, 2, (64, 32), (40, 40, 40), "rk4")


# ---- wide register form (ODE): hidden layers of 65 .. 128 units -- the scripts' argparse default --hidden 128 at x_dim > 16
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,hidden", [(20, 2, (128, 128, 128)), (32, 4, (128, 128)), (17, 0, (100, 90, 70)), (8, 2, (128,)),
                                          (8, 2, (128, 64, 32)), (40, 2, (72, 128, 16))])
def test_wide_register_form_ode(xd, zd, hidden, method):
    check_ode(35, 10, xd, zd, hidden, method)


# ---- LDS form: a DAE with more than four layers, contractions beyond 128 columns, a last layer beyond 64 outputs
@pytest.mark.parametrize("method", ("euler", "rk4"))
@pytest.mark.parametrize("xd,zd,hidden", [(8, 2, (64, 64, 64, 64)),             # depth 4: five Linear layers (register form, 8-layer instance)
                                          (8, 2, (32, 32, 32, 32, 32, 32)),     # seven Linear layers
                                          (8, 2, (128, 128, 128, 128)),         # five layers of 128: beyond both register forms
                                          (70, 10, (64, 64)),                   # 70 outputs: five tiles in the last layer
                                          (40, 10, (96, 96))])                  # 150 columns = 10 quads: beyond the straight-line bodies
def test_deep_and_lds_form_ode(xd, zd, hidden, method):
    check_ode(35, 9, xd, zd, hidden, method)


def test_lds_form_dae():
    check_dae(18, 9, 8, 2, 2, 2, (64, 64, 64, 64), (64, 64, 64, 64), "rk4", combos=((False, False), (True, True)))
    check_dae(18, 9, 20, 10, 40, 40, (64, 64, 64), (64, 64, 64), "midpoint", combos=((False, False), (True, False)))


# ---- streamed form
@pytest.mark.parametrize("method", ("euler", "rk4"))
@pytest.mark.parametrize("xd,zd,hidden", [(20, 2, (160, 160, 160)),     # first layers resident, the last ones streamed
                                          (8, 2, (320, 320, 320)),      # every H -> H layer streamed, 20 tiles = 5 per wave, 20 quads
                                          (8, 2, (200, 72, 200))])      # partial chunks, a wave without a tile in a streamed layer
def test_streamed_form_ode(xd, zd, hidden, method):
    check_ode(33, 8, xd, zd, hidden, method)


def test_streamed_form_dae():
    check_dae(17, 8, 20, 10, 40, 40, (128, 128, 128), (128, 128, 128), "rk4", combos=((False, False), (True, True)))
    check_dae(17, 8, 8, 2, 2, 2, (64, 64, 64), (256, 256, 256), "euler", combos=((False, False),))      # DE in LDS, AE streamed


# ---- the step loop's look-ahead and edge cases
def test_many_external_rows_beyond_the_look_ahead_registers():
    """z + v > 64 rows: the rows beyond 16 per look-ahead register are loaded where they are used."""
    check_dae(9, 7, 4, 40, 45, 3, (64, 64), (64, 64), "rk4", combos=((False, False), (False, True)))
    check_ode(9, 7, 4, 70, (64, 64), "midpoint")


@pytest.mark.parametrize("B,Tn", [(1, 1), (1, 2), (16, 2), (17, 3), (250, 5)])
def test_short_grids_and_ragged_batches(B, Tn):
    check_ode(B, Tn, 20, 2, (64, 64, 64), "rk4")
    check_dae(B, Tn, 8, 4, 6, 6, (64, 64, 64), (64, 64, 64), "rk4", combos=((False, False),))
    check_ode(B, Tn, 8, 2, (64, 64, 64, 64), "euler")


def test_strided_views_and_forced_generic_kernel_agree():
    """The scripts pass permute(1, 0, 2) views; kernel='generic' on a shape that has a specialised integrator is the same function."""
    ls, t, x, z, a0, ev, zj = ode_case(40, 12, 8, 2, (64, 64, 64), 3)
    ref = O.integrate_ode("rk4", ls, t, x, z, a0, ev, zj)
    pv = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)
    out = fused().ode_integrate("rk4", dl(ls), pv(t), pv(x), pv(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel="generic")
    assert rel_err(out.cpu(), ref) <= TOL_GPU
    auto = fused().ode_integrate("rk4", dl(ls), pv(t), pv(x), pv(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda())
    assert rel_err(out.cpu(), auto.cpu()) <= TOL_GPU


def test_full_batch_subset_vs_oracle():
    """4096 x 200 at x_dim 20 (register form): a subset of trajectories against the oracle, the rest finite."""
    B, Tn = 4096, 201
    ls, t, x, z, a0, ev, zj = ode_case(B, Tn, 20, 2, (64, 64, 64), 7)
    out = fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda()).cpu()
    assert torch.isfinite(out).all()
    sel = torch.tensor([0, 1, 15, 16, 17, 2047, 2048, 4079, 4080, 4095])
    ref = O.integrate_ode("rk4", ls, t[:, sel], x[:, sel], z[:, sel], a0[sel], ev[sel], zj[sel])
    assert rel_err(out[:, sel], ref) <= TOL_GPU


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes(seed):
    """Seeded fuzz over layer counts (1 .. 8 Linear layers), widths, state / external dims, batch and grid sizes, method, events and teacher
    forcing: whatever form the launch picks, the oracle decides."""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    method = METHODS[ri(0, 2)]
    nh = ri(0, 7)
    wmax = (24, 64, 64, 130, 200)[ri(0, 4)]
    hidden = tuple(ri(1, wmax) for _ in range(nh))
    B, Tn = ri(1, 50), ri(1, 12)
    if seed % 2 == 0:
        xd, zd = ri(1, 40), ri(0, 30)
        check_ode(B, Tn, xd, zd, hidden, method, seed=seed)
    else:
        xd, zd, vd, idim = ri(1, 24), ri(0, 20), ri(0, 30), ri(1, 20)
        hae = tuple(ri(1, wmax) for _ in range(ri(0, 5)))
        check_dae(B, Tn, xd, zd, vd, idim, hidden[:5], hae, method, seed=seed, combos=((False, False), (True, True), (bool(seed & 2), not (seed & 2))))
