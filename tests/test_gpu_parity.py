"""Parity of the fused HIP path (through the C ABI) against the goldens captured from the reference and
against the CPU oracle.  Tolerance: north_star's "trajectories within 1e-5 rel-err of reference":
per trajectory, max_{t,d} |y - y_ref| / max(max_{t,d} |y_ref|, 1e-3) <= 1e-5  (helpers.traj_rel_err, TOL_GPU)."""
import pytest
import torch

from helpers import TOL_GPU, T, layers, load, tm, traj_rel_err as rel_err
from oracle import psnode_oracle as O

pytestmark = pytest.mark.gpu
METHODS = ("euler", "midpoint", "rk4")
KERNELS = ("generic", "auto")


def dev(a):
    return a.to("cuda") if torch.is_tensor(a) else a


def dl(ls):
    return [(w.cuda(), b.cuda()) for w, b in ls]


def fused():
    from py_psnode_amd import fused as f
    return f


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("method", METHODS)
def test_g2_integrate_ode(method, kernel):
    d = load("g2_ode.npz")
    de = dl(layers(d, "de__x_dot"))
    # the scripts' layout: permuted VIEWS of B-major memory
    t, tr, x, z = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "t_ragged", "x", "z"))
    a0, ev, zj = T(d["all_initial"]).cuda(), T(d["event_t"]).cuda(), T(d["z_jump"]).cuda()
    f = fused()
    run = lambda **kw: f.ode_integrate(method, de, kw.pop("t", t), x, z, a0, kernel=kernel, **kw).cpu()
    assert rel_err(run(event_t=torch.full_like(ev, -1.0), z_jump=zj), d[f"{method}_plain"]) <= TOL_GPU
    assert rel_err(run(), d[f"{method}_noevfn"]) <= TOL_GPU
    assert rel_err(run(event_t=ev, z_jump=zj), d[f"{method}_events"]) <= TOL_GPU
    assert rel_err(run(event_t=ev, z_jump=zj, input_true_x=True), d[f"{method}_events_truex"]) <= TOL_GPU
    assert rel_err(run(t=tr, event_t=ev, z_jump=zj), d[f"{method}_ragged"]) <= TOL_GPU


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("method", METHODS)
def test_g3_integrate_dae(method, kernel):
    d = load("g3_dae.npz")
    de, ae = dl(layers(d, "de__x_dot")), dl(layers(d, "ae__i_calculator"))
    t, x, z, v, i = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "x", "z", "v", "i"))
    xi, a0 = T(d["x_init"]).cuda(), T(d["all_initial"]).cuda()
    ev, zj, vj = T(d["event_t"]).cuda(), T(d["z_jump"]).cuda(), T(d["v_jump"]).cuda()
    f = fused()
    for tx in (False, True):
        for ti in (False, True):
            for use_ev in (False, True):
                kw = dict(event_t=ev, z_jump=zj, v_jump=vj) if use_ev else {}
                xs, is_ = f.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, input_true_x=tx, input_true_i=ti, kernel=kernel, **kw)
                key = f"{method}_tx{int(tx)}_ti{int(ti)}_ev{int(use_ev)}"
                assert rel_err(xs.cpu(), d[key + "_x"]) <= TOL_GPU, key
                assert rel_err(is_.cpu(), d[key + "_i"]) <= TOL_GPU, key
    xe = torch.zeros(x.shape[0], x.shape[1], 0, device="cuda")
    xs, is_ = f.dae_integrate(method, de, ae, xi, t, xe, z, v, i, a0, kernel=kernel)
    assert rel_err(xs.cpu(), d[f"{method}_xdim0_x"]) <= TOL_GPU
    assert rel_err(is_.cpu(), d[f"{method}_xdim0_i"]) <= TOL_GPU


@pytest.mark.parametrize("kernel", ("tile", "wave"))
@pytest.mark.parametrize("method", METHODS)
def test_g2_g5_on_both_mfma_integrators(method, kernel):
    """The reference's goldens G2 (events, teacher forcing, ragged clocks) and G5 (1000 steps) on K1 ("tile") and K1x ("wave") explicitly:
    AUTO picks one of them by batch size, both must hold the goldens."""
    d = load("g2_ode.npz")
    de = dl(layers(d, "de__x_dot"))
    t, tr, x, z = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "t_ragged", "x", "z"))
    a0, ev, zj = T(d["all_initial"]).cuda(), T(d["event_t"]).cuda(), T(d["z_jump"]).cuda()
    f = fused()
    run = lambda **kw: f.ode_integrate(method, de, kw.pop("t", t), x, z, a0, kernel=kernel, **kw).cpu()
    assert rel_err(run(), d[f"{method}_noevfn"]) <= TOL_GPU
    assert rel_err(run(event_t=ev, z_jump=zj), d[f"{method}_events"]) <= TOL_GPU
    assert rel_err(run(event_t=ev, z_jump=zj, input_true_x=True), d[f"{method}_events_truex"]) <= TOL_GPU
    assert rel_err(run(t=tr, event_t=ev, z_jump=zj), d[f"{method}_ragged"]) <= TOL_GPU
    if method != "midpoint":
        d = load("g5_long.npz")
        de = dl(layers(d, "de__x_dot"))
        t, z = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "z"))
        x = torch.zeros(t.shape[0], t.shape[1], 8, device="cuda")
        x[0] = T(d["x0"])[:, 0].cuda()
        assert rel_err(f.ode_integrate(method, de, t, x, z, T(d["all_initial"]).cuda(), kernel=kernel).cpu(), d[method]) <= TOL_GPU


@pytest.mark.parametrize("B,Tn", [(4096, 40), (4609, 12), (5, 300), (1, 2), (130, 3), (64, 66), (7, 130)])
def test_k1_and_k1x_agree_and_auto_picks_by_batch(B, Tn):
    """K1 and K1x on the same call (even and odd x_dim, events crossing the 64-step blocks of the event table, T = 2, T = 3, ragged last
    wave): equal to rounding (plain vs log2e-scaled ELU domain, different summation orders).  AUTO = K1x up to one wave per SIMD
    (B <= 4608), K1 beyond: bit-equal to the forced kernel."""
    for xd, zd, method in ((8, 2, "rk4"), (5, 3, "euler"), (8, 0, "midpoint")):
        ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=xd, zd=zd, seed=3 + B, H=64)
        g = torch.Generator().manual_seed(4)
        ev = zj = None
        if zd and Tn > 4:
            ev = torch.stack([t[1, :, :], t[min(Tn - 2, 65), :, :]], dim=1).contiguous().cuda()
            zj = (0.1 * torch.randn(B, 2, zd, generator=g)).cuda()
        kw = dict(event_t=ev, z_jump=zj)
        f = fused()
        tile = f.ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="tile", **kw)
        wave = f.ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="wave", **kw)
        auto = f.ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="auto", **kw)
        assert rel_err(wave.cpu(), tile.cpu()) <= TOL_GPU
        assert torch.equal(auto, wave if B <= 4608 else tile)


def test_wave_integrators_refuse_views_whose_byte_offsets_do_not_fit_32_bits():
    """K1x / K2x address rows as <uniform row base> + <32-bit per-lane byte offset>.  A view whose batch stride
    spans >= 4 GiB (a B-major dataset of that size handed over as the scripts' permute(1,0,2) view) would wrap: AUTO must fall back to the
    64-bit-indexed K1 and give the same numbers, a forced `wave` must be refused."""
    from py_psnode_amd import _lib
    B, Tn, xd, zd = 6, 9, 8, 2
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=xd, zd=zd, seed=17, H=64)
    f = fused()
    want = f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="tile")
    SB = (1 << 30) // (B - 1) + 8                            # (B - 1) * SB elements * 4 bytes >= 4 GiB
    big = torch.zeros((B - 1) * SB + Tn * zd + 64, device="cuda")
    zv = torch.as_strided(big, (Tn, B, zd), (zd, SB, 1))     # time-major view of "B-major" memory with a huge batch stride
    zv.copy_(z.cuda())
    got = f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), zv, a0.cuda(), kernel="auto")
    assert torch.equal(got, want), "AUTO must take the 64-bit-indexed tile kernel for this view"
    with pytest.raises(_lib.UnsupportedShapeError):
        f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), zv, a0.cuda(), kernel="wave")
    del big, zv
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kernel", ("tile", "wave"))
@pytest.mark.parametrize("method", METHODS)
def test_saving_and_plain_forward_agree(method, kernel):
    """the inference instances run the hidden layers in the log2e-scaled ELU domain, the training forwards (which save the
    rows the backward kernels read) in the plain one -- the same model and inputs under no_grad and under grad agree to rounding, on both
    MFMA integrators, and both with the oracle."""
    B, Tn = 70, 40
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=8, zd=2, seed=29, H=64)
    f = fused()
    plain = f.ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel=kernel)
    saving, saved = f.ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel=kernel, save=True)
    assert saved is not None
    ref = O.integrate_ode(method, ls, t, x, z, a0)
    assert rel_err(plain.cpu(), saving.cpu()) <= 2e-6
    assert rel_err(plain.cpu(), ref) <= TOL_GPU and rel_err(saving.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("kernel", ("tile", "wave"))
@pytest.mark.parametrize("method", METHODS)
def test_g3_on_both_mfma_dae_integrators(method, kernel):
    """The reference's golden G3 (events, x_dim == 0 dataset rows) on K2 ("tile") and K2x ("wave") explicitly."""
    d = load("g3_dae.npz")
    de, ae = dl(layers(d, "de__x_dot")), dl(layers(d, "ae__i_calculator"))
    t, x, z, v, i = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "x", "z", "v", "i"))
    xi, a0 = T(d["x_init"]).cuda(), T(d["all_initial"]).cuda()
    ev, zj, vj = T(d["event_t"]).cuda(), T(d["z_jump"]).cuda(), T(d["v_jump"]).cuda()
    f = fused()
    for evk, kw in (("ev0", {}), ("ev1", dict(event_t=ev, z_jump=zj, v_jump=vj))):
        xs, is_ = f.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, kernel=kernel, **kw)
        assert rel_err(xs.cpu(), d[f"{method}_tx0_ti0_{evk}_x"]) <= TOL_GPU
        assert rel_err(is_.cpu(), d[f"{method}_tx0_ti0_{evk}_i"]) <= TOL_GPU
    xe = torch.zeros(t.shape[0], t.shape[1], 0, device="cuda")
    xs, is_ = f.dae_integrate(method, de, ae, xi, t, xe, z, v, i, a0, kernel=kernel)
    assert rel_err(xs.cpu(), d[f"{method}_xdim0_x"]) <= TOL_GPU
    assert rel_err(is_.cpu(), d[f"{method}_xdim0_i"]) <= TOL_GPU


@pytest.mark.parametrize("kernel", KERNELS)
def test_g5_long_run(kernel):
    d = load("g5_long.npz")
    de = dl(layers(d, "de__x_dot"))
    t, z = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "z"))
    x = torch.zeros(t.shape[0], t.shape[1], 8, device="cuda")
    x[0] = T(d["x0"])[:, 0].cuda()
    a0 = T(d["all_initial"]).cuda()
    f = fused()
    assert rel_err(f.ode_integrate("rk4", de, t, x, z, a0, kernel=kernel).cpu(), d["rk4"]) <= TOL_GPU
    assert rel_err(f.ode_integrate("euler", de, t, x, z, a0, kernel=kernel).cpu(), d["euler"]) <= TOL_GPU


@pytest.mark.parametrize("tag", ["ode01", "ode02", "dae01", "dae02", "dae02_z0"])
@pytest.mark.parametrize("method", METHODS)
def test_g4_models_fused(tag, method):
    """The four scripts' models on the GPU with solver.fused='require': any non-HIP route raises."""
    from py_psnode_amd import neural_dae as nd
    from test_host_models import SOLVERS, _build, _load_sd
    d = load(f"g4_model_{tag}.npz")
    m = _build(tag)
    _load_sd(m, d)
    m = m.cuda()
    m.solver = SOLVERS[method]()
    m.solver.fused = "require"
    g = lambda k: T(d[k]).cuda()
    with torch.no_grad():
        if tag.startswith("ode"):
            out = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        else:
            out = m(t=g("t"), x=g("x"), z=g("z"), v=g("v"), i=g("i"), event_t=g("event_t"), z_jump=g("z_jump"), v_jump=g("v_jump"))
    out = out if isinstance(out, tuple) else (out,)
    for k, o in enumerate(out):
        assert rel_err(o.cpu(), d[f"{method}_out{k}"], bdim=0) <= TOL_GPU, (tag, method, k)
    assert nd.NotFusableError  # imported route check above is the `require` flag


def _synthetic_ode(B, Tn, xd=8, zd=2, H=64, seed=0, n_hidden=3):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    import torch.nn as nn
    dims = [3 * (xd + zd)] + [H] * n_hidden + [xd]
    lin = [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]
    ls = [(l.weight.detach(), l.bias.detach()) for l in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    x = torch.zeros(Tn, B, xd)
    x[0] = 0.1 * torch.randn(B, xd, generator=g)
    z = 0.1 * torch.randn(Tn, B, zd, generator=g)
    a0 = torch.cat((x[0], z[0]), -1)
    return ls, t, x, z, a0


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("B,Tn", [(1, 5), (17, 9), (33, 2), (5, 1), (250, 12)])
def test_ragged_batches_and_short_grids(B, Tn, kernel):
    ls, t, x, z, a0 = _synthetic_ode(B, Tn)
    ref = O.integrate_ode("rk4", ls, t, x, z, a0)
    out = fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel=kernel)
    assert out.shape == ref.shape and out.is_contiguous()
    assert rel_err(out.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("xd,zd,H,nh", [(3, 0, 8, 1), (16, 16, 16, 1), (5, 3, 40, 2), (64, 64, 64, 1)])
def test_other_shapes_generic(xd, zd, H, nh):
    """z_dim == 0, the latent (direct_encode) shapes and odd widths."""
    ls, t, x, z, a0 = _synthetic_ode(6, 7, xd=xd, zd=zd, H=H, n_hidden=nh, seed=3)
    ref = O.integrate_ode("midpoint", ls, t, x, z, a0)
    out = fused().ode_integrate("midpoint", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
    assert rel_err(out.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd", [(8, 2), (3, 0), (5, 3), (8, 4), (1, 1), (8, 0), (8, 6), (4, 8), (8, 7), (12, 2), (16, 4), (9, 0), (13, 8)])
def test_mfma_kernel_shapes(xd, zd, method):
    """Every (x_dim, z_dim) class of the MFMA kernel (NX=2 for x_dim <= 8, NX=4 for x_dim 9..16 -- x_dim is data-defined upstream,
    neural_00_ODE_01_no_encode.py:293; NZM=0..4: z_dim <= 8), forced with kernel='mfma', with events, per-trajectory clocks and a
    ragged last tile."""
    _check_mfma_ode(xd, zd, method, 64)


@pytest.mark.parametrize("H", [32, 100, 128])
@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_mfma_kernel_wide_state_at_other_hidden_widths(method, H):
    """x_dim 9..16 at 2 / 8 waves per tile and zero-padded widths."""
    _check_mfma_ode(12, 2, method, H)
    _check_mfma_ode(16, 4, method, H)


@pytest.mark.parametrize("H", [32, 128])
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd", [(8, 2), (3, 0), (8, 4)])
def test_mfma_kernel_hidden_widths(xd, zd, method, H):
    """--hidden 32 and 128 (the scripts' argparse default, neural_00_ODE_01_no_encode.py:259): 2 and 8 waves per tile."""
    _check_mfma_ode(xd, zd, method, H)


@pytest.mark.parametrize("H", [192, 256, 130, 200])
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd", [(8, 2), (3, 0), (8, 8)])
def test_mfma_kernel_streamed_hidden_widths(xd, zd, method, H):
    """--hidden 129..256 (round 4): 12 / 16 waves per tile, the H->H weights streamed from the L2-resident stream image every layer
    (psnode_mfma_impl.h `weights_streamed`); in-between widths zero-padded to 192 / 256.  Forced with kernel='mfma': events, ragged
    clocks, teacher forcing."""
    _check_mfma_ode(xd, zd, method, H)


@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_mfma_kernel_streamed_widths_wide_state_and_dae(method):
    """At 12 waves (hidden 129..192) the ODE carries x_dim 9..16 and the DAE runs streamed as well (z+v+i <= 6)."""
    _check_mfma_ode(12, 2, method, 192)
    _check_mfma_ode(16, 4, method, 160)
    _check_mfma_dae(8, 2, 2, 2, method, 192)
    _check_mfma_dae(5, 1, 1, 1, method, 144)
    _check_mfma_dae(8, 0, 2, 2, method, 192)


def test_streamed_width_classes_that_fall_back_are_reported_and_still_right():
    """What the streamed widths do NOT carry (they would spill at 128 VGPRs per lane): the DAE and x_dim > 8 above hidden 192, the DAE
    with z+v+i > 6 above 128.  kernel_for says GENERIC, kernel='mfma' refuses, AUTO runs K0 and matches the oracle."""
    import ctypes
    from py_psnode_amd import _lib
    lib = _lib.load()
    a = _lib.OdeArgsF32()
    a.method, a.x_dim, a.z_dim, a.T, a.B = _lib.RK4_38, 8, 2, 11, 16
    a.de.n_layers, a.de.in_dim = 4, 30
    for H, want in ((192, _lib.KERNEL_MFMA), (256, _lib.KERNEL_MFMA), (257, _lib.KERNEL_GENERIC)):
        for k, o in enumerate((H, H, H, 8)):
            a.de.out_dim[k] = o
        assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == want, H
    a.x_dim, a.de.in_dim, a.de.out_dim[3] = 12, 3 * 14, 12
    for H, want in ((192, _lib.KERNEL_MFMA), (256, _lib.KERNEL_GENERIC)):
        for k in range(3):
            a.de.out_dim[k] = H
        assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == want, H
    d = _lib.DaeArgsF32()
    d.method, d.x_dim, d.z_dim, d.v_dim, d.i_dim, d.T, d.B = _lib.RK4_38, 8, 2, 2, 2, 11, 16
    d.de.n_layers, d.de.in_dim, d.ae.n_layers, d.ae.in_dim = 4, 42, 4, 26
    for H, want in ((192, _lib.KERNEL_MFMA), (256, _lib.KERNEL_GENERIC)):
        for k, (o1, o2) in enumerate(zip((H, H, H, 8), (H, H, H, 2))):
            d.de.out_dim[k], d.ae.out_dim[k] = o1, o2
        assert lib.psnode_dae_kernel_for(ctypes.byref(d)) == want, H
    de, ae, t, x, z, v, i, xi, a0, ev, zj, vj = _synthetic_dae(9, 7, 8, 2, 2, 2, seed=23, H=256)
    c = lambda q: q.cuda()
    with pytest.raises(ValueError):
        fused().dae_integrate("rk4", dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), kernel="mfma")
    ref_x, ref_i = O.integrate_dae("rk4", de, ae, xi, t, x, z, v, i, a0)
    xs, is_ = fused().dae_integrate("rk4", dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0))
    assert rel_err(xs.cpu(), ref_x) <= TOL_GPU and rel_err(is_.cpu(), ref_i) <= TOL_GPU


def test_streamed_width_full_batch_subset_vs_oracle():
    """hidden 256 at the headline batch (B=4096 x 200 steps, RK4): 16 waves per tile on every CU, 20 trajectories against the oracle."""
    B, Tn = 4096, 201
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, seed=2, H=256)
    out = fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="mfma")
    idx = torch.tensor(sorted(set(range(0, B, 230)) | {B - 1, 17}))
    ref = O.integrate_ode("rk4", ls, t[:, idx], x[:, idx], z[:, idx], a0[idx])
    assert torch.isfinite(out).all()
    assert rel_err(out[:, idx.cuda()].cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("H", [7, 20, 48, 100, 127])
@pytest.mark.parametrize("method", METHODS)
def test_mfma_kernel_other_hidden_widths_run_zero_padded(method, H):
    """--hidden is a free knob of the scripts: widths between the kernels' 32 / 64 / 128 run on the next one up with zero-padded
    units (zero weights and biases, ELU(0) = 0: exact zeros in every sum) instead of falling to the generic kernel."""
    _check_mfma_ode(8, 2, method, H)
    _check_mfma_dae(8, 2, 2, 2, method, H)


def _check_mfma_ode(xd, zd, method, H):
    B, Tn = 37, 14
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=xd, zd=zd, seed=11, H=H)
    g = torch.Generator().manual_seed(12)
    t = t * (0.5 + torch.rand(1, B, 1, generator=g))
    t[:, 0] = torch.arange(Tn, dtype=torch.float32).view(Tn, 1) * 0.01      # trajectory 0 = the event clock
    ev = torch.stack([t[4, :, :], t[9, :, :]], dim=1).contiguous()           # [B,2,1]
    zj = 0.1 * torch.randn(B, 2, zd, generator=g)
    # hidden <= 64 with x_dim <= 8 has TWO MFMA integrators (round 5): K1 (4-wave tile, "tile") and K1x (one wave per 4 trajectories,
    # "wave"; what "mfma" / "auto" pick at this batch) -- both are checked against the oracle
    kernels = ("tile", "wave") if (H <= 64 and xd <= 8) else ("mfma",)
    ref = O.integrate_ode(method, ls, t, x, z, a0, ev, zj)
    xt = 0.1 * torch.randn(Tn, B, xd, generator=g)
    xt[0] = x[0]
    ref_tf = O.integrate_ode(method, ls, t, xt, z, a0, ev, zj, input_true_x=True)
    for kern in kernels:
        out = fused().ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel=kern)
        assert rel_err(out.cpu(), ref) <= TOL_GPU, kern
        out = fused().ode_integrate(method, dl(ls), t.cuda(), xt.cuda(), z.cuda(), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), input_true_x=True, kernel=kern)
        assert rel_err(out.cpu(), ref_tf) <= TOL_GPU, kern


def _synthetic_dae(B, Tn, xd, zd, vd, idim, seed=0, H=64):
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    n = xd + zd + vd + idim
    mk = lambda dims: [(l.weight.detach(), l.bias.detach()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g)) if B > 1 else t[:, 1:]
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    x, z, v, i, xi = r(Tn, B, xd), r(Tn, B, zd), r(Tn, B, vd), r(Tn, B, idim), r(B, xd)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
    ev = torch.stack([t[3, :, :], t[Tn - 2, :, :]], dim=1).contiguous() if Tn > 5 else None
    return de, ae, t, x, z, v, i, xi, a0, ev, r(B, 2, zd), r(B, 2, vd)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,vd,idim", [(8, 2, 2, 2), (8, 0, 2, 2), (5, 1, 1, 1), (8, 2, 2, 4), (3, 1, 0, 1), (8, 4, 3, 1), (2, 2, 4, 2)])
def test_mfma_dae_kernel_shapes(xd, zd, vd, idim, method):
    """Every (NZM, NZA) class of the DAE MFMA kernel, forced with kernel='mfma': events (incl. the i0 recompute),
    all four teacher-forcing combinations, per-trajectory clocks, ragged last tile."""
    _check_mfma_dae(xd, zd, vd, idim, method, 64)


@pytest.mark.parametrize("H", [32, 128])
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,vd,idim", [(8, 2, 2, 2), (5, 1, 1, 1), (8, 4, 3, 1), (2, 2, 4, 2)])
def test_mfma_dae_kernel_hidden_widths(xd, zd, vd, idim, method, H):
    """--hidden 32 / 128 DAE (at 128 the AE's H->H weights sit in LDS, psnode_mfma_impl.h)."""
    _check_mfma_dae(xd, zd, vd, idim, method, H)


def _check_mfma_dae(xd, zd, vd, idim, method, H):
    B, Tn = 21, 11
    de, ae, t, x, z, v, i, xi, a0, ev, zj, vj = _synthetic_dae(B, Tn, xd, zd, vd, idim, seed=21, H=H)
    c = lambda a: a.cuda()
    for tx in (False, True):
        for ti in (False, True):
            ref_x, ref_i = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, ev, zj, vj, input_true_x=tx, input_true_i=ti)
            # hidden <= 64, i_dim <= 4, no teacher forcing has TWO MFMA integrators (round 5): K2 ("tile") and K2x ("wave", what "mfma" /
            # "auto" pick at this batch); every other call has K2 alone
            kernels = ("tile", "wave") if (H <= 64 and idim <= 4 and not tx and not ti) else ("mfma",)
            for kern in kernels:
                xs, is_ = fused().dae_integrate(method, dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), event_t=c(ev),
                                                z_jump=c(zj), v_jump=c(vj), input_true_x=tx, input_true_i=ti, kernel=kern)
                assert rel_err(xs.cpu(), ref_x) <= TOL_GPU, (tx, ti, kern)
                assert rel_err(is_.cpu(), ref_i) <= TOL_GPU, (tx, ti, kern)


def test_dae_full_size_subset_vs_oracle():
    """BASELINE config 4 (DAE_01 RK4, B=4096, T=1001): 24 trajectories of the full GPU run vs the oracle on those 24."""
    B, Tn = 4096, 1001
    de, ae, t, x, z, v, i, xi, a0, _, _, _ = _synthetic_dae(B, Tn, 8, 2, 2, 2, seed=5)
    c = lambda a: a.cuda()
    xs, is_ = fused().dae_integrate("rk4", dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0))
    idx = torch.tensor(sorted(set(range(0, B, 190)) | {B - 1, 16}))
    ref_x, ref_i = O.integrate_dae("rk4", de, ae, xi[idx], t[:, idx], x[:, idx], z[:, idx], v[:, idx], i[:, idx], a0[idx])
    assert torch.isfinite(xs).all() and torch.isfinite(is_).all()
    assert rel_err(xs[:, idx.cuda()].cpu(), ref_x) <= TOL_GPU
    assert rel_err(is_[:, idx.cuda()].cpu(), ref_i) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
def test_latent_ode_kernel(method):
    """direct_encode ODE latent shape (Linear(6H,H) ELU Linear(H,H), H=16) on the single-wave MFMA kernel, with events,
    per-trajectory clocks, a ragged tile and the scripts' strided views."""
    B, Tn, H = 37, 13, 16
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=H, zd=H, H=H, n_hidden=1, seed=31)
    g = torch.Generator().manual_seed(32)
    t = t * (0.5 + torch.rand(1, B, 1, generator=g))
    t[:, 0] = torch.arange(Tn, dtype=torch.float32).view(Tn, 1) * 0.01
    ev = torch.stack([t[2, :, :], t[9, :, :]], dim=1).contiguous()
    zj = 0.1 * torch.randn(B, 2, H, generator=g)
    ref = O.integrate_ode(method, ls, t, x, z, a0, ev, zj)
    bm = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)      # B-major memory, time-major view
    out = fused().ode_integrate(method, dl(ls), bm(t), bm(x), bm(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel="mfma")
    assert rel_err(out.cpu(), ref) <= TOL_GPU
    gen = fused().ode_integrate(method, dl(ls), bm(t), bm(x), bm(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel="generic")
    assert rel_err(gen.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("zd", [16, 0])
def test_latent_dae_kernel(method, zd):
    """direct_encode DAE latent shapes (12H|9H -> H -> H and 7H|5H -> H -> H, H=16), with and without z."""
    import torch.nn as nn
    B, Tn, H = 21, 9, 16
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(41)
    nblk = 4 if zd else 3
    mk = lambda dims: [(l.weight.detach(), l.bias.detach()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    de, ae = mk([3 * nblk * H, H, H]), mk([(2 * nblk - 1) * H, H, H])
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    x, z, v, i, xi = r(Tn, B, H), r(Tn, B, zd), r(Tn, B, H), r(Tn, B, H), r(B, H)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
    ev = torch.stack([t[3, :, :], t[6, :, :]], dim=1).contiguous()
    zj, vj = r(B, 2, zd), r(B, 2, H)
    ref_x, ref_i = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, ev, zj, vj)
    c = lambda a: a.cuda()
    xs, is_ = fused().dae_integrate(method, dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), event_t=c(ev), z_jump=c(zj),
                                    v_jump=c(vj), kernel="mfma")
    assert rel_err(xs.cpu(), ref_x) <= TOL_GPU and rel_err(is_.cpu(), ref_i) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
def test_latent64_ode_kernel(method):
    """direct_encode ODE latent shape with hidden 64 (384 -> 64 -> 64) on K3c (kernel='mfma') vs the oracle."""
    B, Tn, H = 21, 11, 64
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, xd=H, zd=H, H=H, n_hidden=1, seed=51)
    g = torch.Generator().manual_seed(52)
    t = t * (0.5 + torch.rand(1, B, 1, generator=g))
    t[:, 0] = torch.arange(Tn, dtype=torch.float32).view(Tn, 1) * 0.01
    ev = torch.stack([t[2, :, :], t[8, :, :]], dim=1).contiguous()
    zj = 0.1 * torch.randn(B, 2, H, generator=g)
    ref = O.integrate_ode(method, ls, t, x, z, a0, ev, zj)
    bm = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)
    out = fused().ode_integrate(method, dl(ls), bm(t), bm(x), bm(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel="mfma")
    assert rel_err(out.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("zd", [64, 0])
def test_latent64_dae_kernel(method, zd):
    """direct_encode DAE latent shapes with hidden 64 (768|576 -> 64 -> 64 and 448|320 -> 64 -> 64): the shipped DAE_02 config."""
    import torch.nn as nn
    B, Tn, H = 19, 8, 64
    g = torch.Generator().manual_seed(61)
    torch.manual_seed(61)
    nblk = 4 if zd else 3
    mk = lambda dims: [(l.weight.detach(), l.bias.detach()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    de, ae = mk([3 * nblk * H, H, H]), mk([(2 * nblk - 1) * H, H, H])
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    x, z, v, i, xi = r(Tn, B, H), r(Tn, B, zd), r(Tn, B, H), r(Tn, B, H), r(B, H)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
    ev = torch.stack([t[3, :, :], t[Tn - 2, :, :]], dim=1).contiguous()      # an event on the last step too
    zj, vj = r(B, 2, zd), r(B, 2, H)
    ref_x, ref_i = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, ev, zj, vj)
    c = lambda a: a.cuda()
    xs, is_ = fused().dae_integrate(method, dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), event_t=c(ev), z_jump=c(zj),
                                    v_jump=c(vj), kernel="mfma")
    assert rel_err(xs.cpu(), ref_x) <= TOL_GPU and rel_err(is_.cpu(), ref_i) <= TOL_GPU


@pytest.mark.parametrize("din,dout", [(8, 16), (2, 16), (16, 8), (16, 2), (16, 16), (3, 5), (6, 16)])
@pytest.mark.parametrize("rows", [1, 17, 4100])
def test_row_mlp_kernel(din, dout, rows):
    """Encoder / decoder row kernel vs the nn.Sequential it replaces (fp32 reference on CPU)."""
    import torch.nn as nn
    torch.manual_seed(din * 100 + dout)
    _row_mlp_case(din, 16, dout, rows)


@pytest.mark.parametrize("din,dout", [(8, 64), (2, 64), (64, 8), (64, 2), (64, 64), (5, 64)])
@pytest.mark.parametrize("rows", [17, 4100])
def test_row_mlp_kernel_hidden64(din, dout, rows):
    """Hidden 64 (the shipped DAE_02 config): encoders in->64->64, decoders 64->64->out."""
    _row_mlp_case(din, 64, dout, rows)


def _row_mlp_case(din, H, dout, rows):
    import torch.nn as nn
    torch.manual_seed(din * 100 + dout)
    seq = nn.Sequential(nn.Linear(din, H), nn.ELU(), nn.Linear(H, dout))
    inp = torch.randn(rows, din)
    with torch.no_grad():
        ref = seq(inp)
    ls = [(seq[0].weight.detach().cuda(), seq[0].bias.detach().cuda()), (seq[2].weight.detach().cuda(), seq[2].bias.detach().cuda())]
    out = fused().mlp_rows(ls, inp.cuda())
    assert out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    # strided 3-D input like the scripts' [B,T,D] tensors
    if rows == 4100:
        inp3 = torch.randn(41, 100, din)
        with torch.no_grad():
            ref3 = seq(inp3)
        out3 = fused().mlp_rows(ls, inp3.cuda())
        assert out3.shape == ref3.shape and float((out3.cpu() - ref3).abs().max()) <= 2e-6 * max(1.0, float(ref3.abs().max()))


def test_auto_picks_mfma_for_reference_shape():
    import ctypes
    from py_psnode_amd import _lib
    lib = _lib.load()
    a = _lib.OdeArgsF32()
    a.method, a.x_dim, a.z_dim, a.T, a.B = _lib.RK4_38, 8, 2, 1001, 4096
    a.de.n_layers, a.de.in_dim = 4, 30
    for k, o in enumerate((64, 64, 64, 8)):
        a.de.out_dim[k] = o
    assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == _lib.KERNEL_MFMA_WAVE      # K1x: hidden <= 64 at up to one wave per SIMD (B <= 4608)
    a.B = 32768
    assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == _lib.KERNEL_MFMA           # the node-size batch: K1 (two and more tiles per CU)
    a.B = 4096
    a.de.out_dim[1] = 32                       # mixed widths: no MFMA kernel
    assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == _lib.KERNEL_GENERIC
    for H, want in ((32, _lib.KERNEL_MFMA_WAVE), (128, _lib.KERNEL_MFMA)):      # --hidden 32 runs zero-padded on K1x, 128 on K1's 8-wave instance
        for k in range(3):
            a.de.out_dim[k] = H
        assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == want
    d = _lib.DaeArgsF32()
    d.method, d.x_dim, d.z_dim, d.v_dim, d.i_dim, d.T, d.B = _lib.RK4_38, 8, 2, 2, 2, 1001, 4096
    d.de.n_layers, d.de.in_dim, d.ae.n_layers, d.ae.in_dim = 4, 42, 4, 26
    for k, (o1, o2) in enumerate(zip((64, 64, 64, 8), (64, 64, 64, 2))):
        d.de.out_dim[k], d.ae.out_dim[k] = o1, o2
    assert lib.psnode_dae_kernel_for(ctypes.byref(d)) == _lib.KERNEL_MFMA_WAVE       # K2x: as K1x, i_dim <= 4, no teacher forcing
    d.B = 32768
    assert lib.psnode_dae_kernel_for(ctypes.byref(d)) == _lib.KERNEL_MFMA
    d.B = 4096
    for H, want in ((48, _lib.KERNEL_MFMA_WAVE), (100, _lib.KERNEL_MFMA)):      # in-between widths run zero-padded on the next instantiation up
        for k in range(3):
            a.de.out_dim[k] = H
        assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == want
    for xd_ in (9, 12, 16):                    # x_dim 9..16 (data-defined upstream): four x registers per lane, still MFMA
        a.x_dim, a.de.in_dim, a.de.out_dim[3] = xd_, 3 * (xd_ + 2), xd_
        assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == _lib.KERNEL_MFMA
    a.x_dim, a.de.in_dim, a.de.out_dim[3] = 17, 3 * 19, 17
    assert lib.psnode_ode_kernel_for(ctypes.byref(a)) == _lib.KERNEL_GENERIC
    ls, t, x, z, a0 = _synthetic_ode(4, 3, H=300)
    with pytest.raises(ValueError):      # PSNODE_ERR_UNSUPPORTED: no MFMA kernel above hidden 256
        fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="mfma")


def test_full_size_subset_vs_oracle():
    """BASELINE config 2 (B=4096, T=1001, RK4): trajectories are independent, so 48 of them taken from the
    full GPU run must match the oracle run on just those 48."""
    B, Tn = 4096, 1001
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, seed=1)
    out = fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
    torch.cuda.synchronize()
    idx = torch.tensor(sorted(set(range(0, B, 97)) | {B - 1, 15, 16, 17}))
    ref = O.integrate_ode("rk4", ls, t[:, idx], x[:, idx], z[:, idx], a0[idx])
    sub = out[:, idx.cuda()].cpu()
    assert torch.isfinite(out).all()
    assert rel_err(sub, ref) <= TOL_GPU


def test_config5_node_batch_on_one_gpu():
    """BASELINE config 5's GLOBAL batch (32 768 trajectories = 8 tiles per CU: the regime where two waves share a SIMD) on one GPU,
    1000 RK4 steps: 40 trajectories spread over the batch against the oracle run on just those, and -- trajectories being independent --
    rank r's 4096-trajectory shard integrated on its own BY THE SAME KERNEL must reproduce its slice of the full run BIT FOR BIT (the gloo
    test checks the gather, this checks that a shard does not depend on who shares the launch).  Round 5: AUTO picks the integrator by the
    launch's batch (K1x up to one wave per SIMD = the 4096-trajectory shard, K1 for the node-size launch): the shard as the 8-GPU run
    integrates it agrees with the node-size launch to rounding, and is itself checked against the oracle."""
    B, Tn = 32768, 1001
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, seed=5)
    lay = dl(ls)
    tc, xc, zc, ac = t.cuda(), x.cuda(), z.cuda(), a0.cuda()
    out = fused().ode_integrate("rk4", lay, tc, xc, zc, ac)
    assert torch.isfinite(out).all()
    idx = torch.tensor(sorted(set(range(0, B, 911)) | {B - 1, 4095, 4096, 16383}))
    ref = O.integrate_ode("rk4", ls, t[:, idx], x[:, idx], z[:, idx], a0[idx])
    assert rel_err(out[:, idx.cuda()].cpu(), ref) <= TOL_GPU
    for r in (0, 3, 7):
        lo, hi = 4096 * r, 4096 * (r + 1)
        shard = fused().ode_integrate("rk4", lay, tc[:, lo:hi], xc[:, lo:hi], zc[:, lo:hi], ac[lo:hi], kernel="tile")
        assert torch.equal(shard, out[:, lo:hi]), f"shard {r} differs from its slice of the node-size launch"
        auto = fused().ode_integrate("rk4", lay, tc[:, lo:hi], xc[:, lo:hi], zc[:, lo:hi], ac[lo:hi])      # K1x, as on the 8-GPU node
        assert rel_err(auto.cpu(), out[:, lo:hi].cpu()) <= TOL_GPU
        sel = idx[(idx >= lo) & (idx < hi)]
        assert rel_err(auto[:, (sel - lo).cuda()].cpu(), O.integrate_ode("rk4", ls, t[:, sel], x[:, sel], z[:, sel], a0[sel])) <= TOL_GPU


def test_accuracy_equivalent_to_reference_vs_fp64():
    """Against an fp64 evaluation of the same algorithm, the HIP result must be as accurate as the reference's own
    fp32 result (goldens), up to 3x, under BOTH metrics -- on the 1000-step run where roundoff accumulates most."""
    from helpers import rel_err as elem_err
    d = load("g5_long.npz")
    de = layers(d, "de__x_dot")
    t, z = tm(d["t"]), tm(d["z"])
    x = torch.zeros(t.shape[0], t.shape[1], 8)
    x[0] = T(d["x0"])[:, 0]
    a0 = T(d["all_initial"])
    D = lambda a: a.double()
    truth = O.integrate_ode("rk4", [(D(w), D(b)) for w, b in de], D(t), D(x), D(z), D(a0))
    out = fused().ode_integrate("rk4", dl(de), t.cuda(), x.cuda(), z.cuda(), a0.cuda()).cpu()
    assert rel_err(out, truth) <= 3 * rel_err(T(d["rk4"]), truth) + 1e-7
    assert elem_err(out, truth) <= 3 * elem_err(T(d["rk4"]), truth) + 1e-7


def test_time_chunked_launches_are_bit_identical():
    """The multi-GPU pipeline integrates in time chunks that restart from the previous chunk's last row; that must
    reproduce the single launch bit for bit (ODE with events, and DAE where the restart recomputes i0)."""
    from py_psnode_amd import sharded
    d = load("g2_ode.npz")
    de = dl(layers(d, "de__x_dot"))
    t, x, z = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "x", "z"))
    a0, ev, zj = T(d["all_initial"]).cuda(), T(d["event_t"]).cuda(), T(d["z_jump"]).cuda()
    f = fused()
    tab = f.event_table(t, ev)
    one = f.ode_integrate("rk4", de, t, x, z, a0, event_idx=tab, z_jump=zj)
    xs, _ = sharded.integrate_ode_pipelined("rk4", de, t, x, z, a0, event_idx=tab, z_jump=zj, chunks=5, gather=False)
    assert torch.equal(xs, one)
    d = load("g3_dae.npz")
    de, ae = dl(layers(d, "de__x_dot")), dl(layers(d, "ae__i_calculator"))
    t, x, z, v, i = (T(d[k]).cuda().permute(1, 0, 2) for k in ("t", "x", "z", "v", "i"))
    xi, a0 = T(d["x_init"]).cuda(), T(d["all_initial"]).cuda()
    one_x, one_i = f.dae_integrate("rk4", de, ae, xi, t, x, z, v, i, a0)
    Tn, B = t.shape[0], t.shape[1]
    xs, is_ = torch.empty(Tn, B, 8, device="cuda"), torch.empty(Tn, B, 2, device="cuda")
    bnd = sharded.chunk_bounds(Tn, 3)
    for c in range(3):
        s, r1 = max(bnd[c] - 1, 0), bnd[c + 1]
        f.dae_integrate("rk4", de, ae, xi if c == 0 else xs[s], t[s:r1], x[s:r1], z[s:r1], v[s:r1], i[s:r1], a0, out=(xs[s:r1], is_[s:r1]))
    assert torch.equal(xs, one_x) and torch.equal(is_, one_i)


def test_batch_permutation_equivariance_full_size():
    B, Tn = 4096, 201
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, seed=2)
    f = fused()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    a = f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
    b = f.ode_integrate("rk4", dl(ls), t[:, perm].cuda(), x[:, perm].cuda(), z[:, perm].cuda(), a0[perm].cuda())
    assert torch.equal(a[:, perm.cuda()], b)          # bit-exact: a trajectory's arithmetic does not depend on its slot


def test_event_table_kernel_matches_reference_semantics():
    d = load("g2_ode.npz")
    t, ev = T(d["t"]).cuda().permute(1, 0, 2), T(d["event_t"]).cuda()
    tab = fused().event_table(t, ev).cpu().tolist()
    assert tab == O.event_step_table(tm(d["t"]), T(d["event_t"]))
    dup = ev.clone()
    dup[:, 1] = dup[:, 0]
    with pytest.raises(RuntimeError):
        fused().event_table(t, dup, check_duplicates=True)


def test_duplicate_event_check_is_memoised_per_tensor_version():
    """The duplicate-event check reads 4 bytes back (a device sync); for the same clock / event tensors -- or fresh VIEWS of them, as
    the solver route makes on every call -- at the same in-place version it is done once.  An in-place edit is seen."""
    f = fused()
    d = load("g2_ode.npz")
    t_bm, ev = T(d["t"]).cuda(), T(d["event_t"]).cuda()       # B-major clock as the scripts hold it; the solver permutes per call
    f._DUP_OK.clear()
    assert not f._dup_check_known(t_bm.permute(1, 0, 2), ev)
    tab = f.event_table(t_bm.permute(1, 0, 2), ev, check_duplicates=True)
    assert f._dup_check_known(t_bm.permute(1, 0, 2), ev)      # a NEW view object of the same base: known
    assert torch.equal(tab, f.event_table(t_bm.permute(1, 0, 2), ev, check_duplicates=True))
    ev[:, 1] = ev[:, 0]                                       # in-place edit: version bump, checked again, and now it raises
    assert not f._dup_check_known(t_bm.permute(1, 0, 2), ev)
    with pytest.raises(RuntimeError):
        f.event_table(t_bm.permute(1, 0, 2), ev, check_duplicates=True)


def test_c_abi_error_codes_on_device():
    f = fused()
    ls, t, x, z, a0 = _synthetic_ode(4, 3)
    with pytest.raises(ValueError):
        f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda()[:, :9])
    with pytest.raises(TypeError):
        f.ode_integrate("rk4", dl(ls), t.cuda(), x.cuda().double(), z.cuda(), a0.cuda())
    bad = dl(ls)
    bad[0] = (bad[0][0][:, :29].contiguous(), bad[0][1])
    with pytest.raises(ValueError):
        f.ode_integrate("rk4", bad, t.cuda(), x.cuda(), z.cuda(), a0.cuda())


def test_order_of_accuracy_full_batch():
    """Size-independent property: against a fine-grid fp64 solution the 3/8-rule's global error must
    fall by ~2^4 when h halves (13x-17x here: ELU is only C^1 at 0) and Euler's by ~2.  The GPU runs the full config-2
    batch (B=4096); the fp64 truth is computed by the oracle for 32 of those trajectories."""
    import torch.nn as nn
    torch.manual_seed(0)
    xd, zd, H, B, Tend = 8, 2, 64, 4096, 4.0
    n = xd + zd
    lin = [nn.Linear(d0, d1) for d0, d1 in zip([3 * n, H, H, H], [H, H, H, xd])]
    with torch.no_grad():
        for l in lin:
            l.weight.mul_(1.5)                     # livelier dynamics: truncation error well above fp32 roundoff
    ls = [(l.weight.detach(), l.bias.detach()) for l in lin]
    g = torch.Generator().manual_seed(1)
    x0, zc = 0.5 * torch.randn(B, xd, generator=g), 0.5 * torch.randn(B, zd, generator=g)
    a0 = torch.cat((x0, zc), -1)
    idx = torch.arange(0, B, B // 32)

    def grid(steps, dt, sel=slice(None)):
        t = (torch.arange(steps + 1, dtype=dt) * (Tend / steps)).view(-1, 1, 1).repeat(1, x0[sel].shape[0], 1)
        x = torch.zeros(steps + 1, x0[sel].shape[0], xd, dtype=dt)
        x[0] = x0[sel].to(dt)
        return t, x, zc[sel].to(dt).unsqueeze(0).repeat(steps + 1, 1, 1)

    t, x, z = grid(4096, torch.float64, idx)
    truth = O.integrate_ode("rk4", [(w.double(), b.double()) for w, b in ls], t, x, z, a0[idx].double())[-1]
    ratios = {}
    for method in ("euler", "rk4"):
        errs = []
        # (RK4 on 2 / 4 / 8 steps: at 16 its truncation error, 1e-6, is the fp32 roundoff of the integration itself)
        for steps in ((4, 8, 16) if method == "euler" else (2, 4, 8)):
            t, x, z = grid(steps, torch.float32)
            out = fused().ode_integrate(method, dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
            errs.append(float((out[-1][idx.cuda()].double().cpu() - truth).abs().max()))
        ratios[method] = [errs[k] / errs[k + 1] for k in range(2)]
    assert all(8.0 < r < 24.0 for r in ratios["rk4"]), (ratios, errs)
    assert all(1.8 < r < 2.2 for r in ratios["euler"]), ratios


def test_dae02_shipped_config_hidden64_runs_fused():
    """neural_01_DAE_02_direct_encode.py ships with hidden = 64 (its debug override, :267): latent blocks of 64, DE 768->64->64,
    AE 448->64->64.  No MFMA specialisation -> generic kernel; must still be a HIP launch ('require') and match the model
    evaluated by the callback walk on the CPU."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(5)
    B, Tn = 19, 9
    m = models.DAE_Model(8, 2, 2, 2, 64, direct_encode=True, solver=nd.RK4())
    g = torch.Generator().manual_seed(6)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    x, z, v, i = r(B, Tn, 8), r(B, Tn, 2), r(B, Tn, 2), r(B, Tn, 2)
    ev, zj, vj = t[:, [3, 6], :].contiguous(), r(B, 2, 2), r(B, 2, 2)
    with torch.no_grad():
        ref = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
        mg = m.cuda()
        mg.solver.fused = "require"
        c = lambda a: a.cuda()
        out = mg(t=c(t), x=c(x), z=c(z), v=c(v), i=c(i), event_t=c(ev), z_jump=c(zj), v_jump=c(vj))
    for k, (o, rr) in enumerate(zip(out, ref)):
        assert rel_err(o.cpu(), rr, bdim=0) <= TOL_GPU, k


def test_shapes_beyond_every_kernel_fall_back_to_the_walk():
    """An MLP too wide for the generic kernel's LDS budget: 'auto' walks the user's modules, 'require' raises."""
    from py_psnode_amd import _lib, models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(1)
    de = models.DE_Func(700, (64,), 350).cuda()          # in_features 2100 > PSNODE_MAX_IN_WIDTH
    B, Tn = 3, 3
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    x, z = 0.1 * torch.randn(Tn, B, 350).cuda(), 0.1 * torch.randn(Tn, B, 350).cuda()
    a0 = torch.cat((x[0], z[0]), -1)
    s = nd.Euler()
    with torch.no_grad():
        with pytest.warns(RuntimeWarning, match="not fusable"):          # the walk on a HIP device is never silent
            out = s.integrate_ODE(x_func=de, t=t, x=x, z=z, all_initial=a0)
        assert out.shape == x.shape and torch.isfinite(out).all()
        s.fused = "require"
        with pytest.raises((_lib.UnsupportedShapeError, ValueError)):
            s.integrate_ODE(x_func=de, t=t, x=x, z=z, all_initial=a0)


def test_fused_call_is_capturable_in_a_hip_graph():
    """Nothing in a fused call synchronises the host, so the 3-launch sequence (pack, event table, integrator) can be captured in a
    HIP graph by the caller and replayed on new data in the same buffers."""
    B, Tn = 64, 40
    ls, t, x, z, a0 = _synthetic_ode(B, Tn, seed=77)
    layers = dl(ls)
    tc, xc, zc, ac = t.cuda(), x.cuda(), z.cuda(), a0.cuda()
    out = torch.empty(Tn, B, 8, device="cuda")
    fused().ode_integrate("rk4", layers, tc, xc, zc, ac, out=out)            # warm-up outside the capture (workspace sizing etc.)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            fused().ode_integrate("rk4", layers, tc, xc, zc, ac, out=out)
    ls2, t2, x2, z2, a02 = _synthetic_ode(B, Tn, seed=78)
    xc.copy_(x2.cuda()); zc.copy_(z2.cuda()); ac.copy_(a02.cuda())
    graph.replay()
    torch.cuda.synchronize()
    ref = fused().ode_integrate("rk4", layers, tc, xc, zc, ac)
    assert torch.equal(out, ref)
    assert rel_err(out.cpu(), O.integrate_ode("rk4", ls, t, x2, z2, a02)) <= TOL_GPU


def test_inline_elu_accuracy_contract():
    """The MFMA kernels' inline ELU (psnode_common.h:elu_pair) seen through the row kernel with identity weights
    (Linear(16,16)=I, ELU, Linear(16,16)=I: fp32 MFMA products with 1.0 and sums with 0.0 are exact, so out == ELU(in) bit for bit).
    Contract since round 3 (max(x,0) + (exp2(min(x,0) log2e) - 1), DESIGN.md "ELU"): exact identity for x > 0; ABSOLUTE error vs fp64
    expm1 <= 1.2e-7 over the whole negative range (2 ulp at 1.0); a few ulp RELATIVE from -0.25 down; exact 0 at 0 and for arguments
    too small to move exp2 off 1.0; saturates at -1.  (Rounds 1-2 additionally kept relative accuracy for x -> 0-, as ATen's expm1
    does: -DPSNODE_ELU_EXPM1 builds that form; the trajectories are no closer to the reference with it, profiles/r03b_elu_exp2_ab.txt.)"""
    eye = torch.eye(16)
    ls = [(eye.cuda(), torch.zeros(16).cuda()), (eye.cuda(), torch.zeros(16).cuda())]
    mags = torch.cat((torch.logspace(-38, 2, 16 * 4096, dtype=torch.float64), torch.linspace(0.2, 0.3, 16 * 256, dtype=torch.float64),
                      torch.tensor([0.25, 0.2500001, 0.2499999, 17.0, 88.0, 104.0, 1e4, 0.0] * 2, dtype=torch.float64)))
    x = torch.cat((-mags, mags)).float()
    out = fused().mlp_rows(ls, x.view(-1, 16).cuda()).cpu().view(-1).double()
    xd = x.double()
    ref = torch.where(xd > 0, xd, torch.expm1(xd))
    pos = xd > 0
    assert torch.equal(out[pos], xd[pos]), "ELU(x) must be exactly x for x > 0"
    neg = ~pos & (xd != 0)
    err = (out[neg] - ref[neg]).abs()
    print(f"inline ELU: max absolute error vs fp64 expm1 {float(err.max()):.3e} at x = {float(xd[neg][err.argmax()]):.6g}")
    assert float(err.max()) <= 1.2e-7
    deep = neg & (xd <= -0.25)
    rel = (out[deep] - ref[deep]).abs() / ref[deep].abs()
    assert float(rel.max()) <= 4e-7, float(rel.max())
    assert float(out[xd == 0].abs().max()) == 0.0
    assert float(out[neg].max()) <= 0.0 and float(out[neg].min()) >= -1.0
    assert float(out[neg & (xd.abs() < 1e-9)].abs().max()) == 0.0
    assert float((out[xd <= -88.0] + 1.0).abs().max()) == 0.0


def test_module_with_another_forward_is_walked_not_fused():
    """A DE_Func-shaped module whose forward() is NOT the recipe: 'auto' steps through the user's callable (correct result),
    'require' raises -- it is never integrated with the hard-coded recipe."""
    import warnings
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd

    class Scaled(models.DE_Func):
        def forward(self, t0, xt, zt, all_initial):
            return 0.5 * super().forward(t0, xt, zt, all_initial)

    torch.manual_seed(3)
    de = Scaled(10, (64, 64, 64), 8).cuda()
    B, Tn = 9, 7
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    x, z = (0.1 * torch.randn(Tn, B, 8)).cuda(), (0.1 * torch.randn(Tn, B, 2)).cuda()
    a0 = torch.cat((x[0], z[0]), -1)
    s = nd.RK4()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = s.integrate_ODE(x_func=de, t=t, x=x, z=z, all_initial=a0)
        ls = [(w.detach().cpu(), b.detach().cpu()) for w, b in fused().sequential_layers(de.x_dot)]
        half = [(w, b) for w, b in ls[:-1]] + [(0.5 * ls[-1][0], 0.5 * ls[-1][1])]       # 0.5 * MLP == MLP with the last layer halved
        ref = O.integrate_ode("rk4", half, t.cpu(), x.cpu(), z.cpu(), a0.cpu())
    assert rel_err(out.cpu(), ref) <= TOL_GPU
    s.fused = "require"
    with torch.no_grad(), pytest.raises(nd.NotFusableError):
        s.integrate_ODE(x_func=de, t=t, x=x, z=z, all_initial=a0)


def test_raw_entry_points_validate_leading_dims():
    """mismatched z / t / jump shapes must be a ValueError, not an out-of-bounds device read; a float64 clock on the
    solver route means 'walk', not a TypeError."""
    import warnings
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    ls, t, x, z, a0 = _synthetic_ode(8, 6, seed=3)
    c = lambda a: a.cuda()
    f = fused()
    with pytest.raises(ValueError):
        f.ode_integrate("rk4", dl(ls), c(t), c(x), c(z[:, :5]), c(a0))                  # z covers 5 of 8 trajectories
    with pytest.raises(ValueError):
        f.ode_integrate("rk4", dl(ls), c(t), c(x), c(z[:4]), c(a0))                     # z covers 4 of 6 grid points
    ev = t[2, :, :].unsqueeze(1).contiguous()
    with pytest.raises(ValueError):
        f.ode_integrate("rk4", dl(ls), c(t), c(x), c(z), c(a0), event_t=c(ev), z_jump=torch.zeros(5, 1, 2).cuda())
    torch.manual_seed(0)
    m = models.ODE_Model(8, 2, 64, solver=nd.RK4()).cuda()
    bt = lambda a: a.permute(1, 0, 2).contiguous().cuda()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            m(t=bt(t).double(), x=bt(x), z=bt(z), event_t=torch.full((8, 1, 1), -1.0).cuda().double(), z_jump=torch.zeros(8, 1, 2).cuda())
        except RuntimeError:
            pass                                     # the walk itself may reject mixed dtypes in torch ops: that is the reference's behaviour
