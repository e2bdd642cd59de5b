"""Row f1 pinned to the REFERENCE: gradients that the reference's own `loss.backward()` produced through integrate_ODE /
integrate_DAE (tests/golden/make_goldens_r2.py ran the four scripts' ODE_Model / DAE_Model from /root/reference and stored
every parameter .grad and the input gradients, golden set G7).

  CPU  (not gpu): this package's callback walk under fp32 autograd reproduces them (pins the host mirror's training path).
  GPU  (-m gpu) : the fused route -- K1/K2/K3a/K3c forward, K4/K7/K8/K9 (+ row-MLP backward) -- and, through the raw-tensor
                  API, the generic backward K5, against the same arrays.  Tolerance: 1e-5 of each tensor's max magnitude
                  (round 5: ~6 x the worst achieved error, profiles/r05_grad_accuracy_report.txt).
"""
import pytest
import torch

from helpers import T, layers, load, tm
from py_psnode_amd import models
from py_psnode_amd import neural_dae as nd

SOLVERS = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}
TAGS = ["ode01", "dae01", "ode02", "ode02_h64", "dae02", "dae02_h64", "dae02_z0",
        # round 3 (make_goldens_r3.py): the scripts' argparse default --hidden 128 and --hidden 32 -- K4f (ODE) / K7w (DAE) on the GPU
        "ode01_h128", "ode01_h32", "dae01_h128", "dae01_h32"]
TOL_CPU = 2e-5      # same ATen ops in (almost) the same order as the reference
TOL_GPU = 1e-5      # ~6 x the worst achieved error over every model x method x tensor (1.63e-6: profiles/r05_grad_accuracy_report.txt; the reference's
                    # own fp32 gradients sit 1.2e-6 from an fp64 walk)


def _build(tag):
    if tag.startswith("ode01"):
        return models.ODE_Model(8, 2, int(tag.split("_h")[1]) if "_h" in tag else 64)
    if tag.startswith("dae01"):
        return models.DAE_Model(8, 2, 2, 2, int(tag.split("_h")[1]) if "_h" in tag else 64)
    if tag.startswith("ode02"):
        return models.ODE_Model(8, 2, 64 if tag.endswith("h64") else 16, direct_encode=True)
    return models.DAE_Model(8, 0 if tag.endswith("z0") else 2, 2, 2, 64 if tag.endswith("h64") else 16, direct_encode=True)


def _close(a, b, what, tol):
    b = torch.as_tensor(b, dtype=torch.float64)
    a = a.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(b.abs().max()) if b.numel() else 0.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * max(scale, 1e-6), f"{what}: err {err:.3e} vs scale {scale:.3e}"


def _run_model(tag, method, dev, fused_mode):
    d = load(f"g7_grad_{tag}.npz")
    m = _build(tag)
    sd = {k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(dev)
    m.solver = SOLVERS[method]()
    m.solver.fused = fused_mode
    c = lambda k: T(d[k]).to(dev)
    leaves = {k: c(k).requires_grad_(True) for k in ("x", "z", "v", "i", "z_jump", "v_jump")}
    if tag.startswith("dae"):
        res = m(t=c("t"), x=leaves["x"], z=leaves["z"], v=leaves["v"], i=leaves["i"], event_t=c("event_t"),
                z_jump=leaves["z_jump"], v_jump=leaves["v_jump"])
    else:
        res = m(t=c("t"), x=leaves["x"], z=leaves["z"], event_t=c("event_t"), z_jump=leaves["z_jump"])
    res = res if isinstance(res, tuple) else (res,)
    sum((r * c(f"G{k}")).sum() for k, r in enumerate(res)).backward()
    return d, m, res, leaves


def _check(tag, method, d, m, res, leaves, tol):
    for k, r in enumerate(res):
        _close(r, d[f"{method}_out{k}"], f"{tag} {method} out{k}", tol)
    for name, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        _close(g, d[f"{method}_gp__" + name.replace(".", "__")], f"{tag} {method} grad {name}", tol)
    for k, a in leaves.items():
        key = f"{method}_g_{k}"
        if key in d:
            g = a.grad if a.grad is not None else torch.zeros_like(a)
            _close(g, d[key], f"{tag} {method} grad {k}", tol)
        elif a.numel():
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, f"{tag} {method}: reference leaves {k} without gradient"


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_walk_backward_matches_reference_gradients(tag, method):
    d, m, res, leaves = _run_model(tag, method, "cpu", "auto")
    _check(tag, method, d, m, res, leaves, TOL_CPU)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_fused_backward_matches_reference_gradients(tag, method):
    """solver.fused = 'require': the fused forward + fused backward kernels must take the call (K4 ode01, K7 dae01, K8 ode02,
    K9 ode02_h64 / dae02_h64, the single-wave latent DAE backward for dae02 / dae02_z0; encoders/decoders on the row kernels)."""
    d, m, res, leaves = _run_model(tag, method, "cuda", "require")
    _check(tag, method, d, m, res, leaves, TOL_GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ode01", "dae01", "ode02_h64", "dae02_h64", "ode01_h128", "dae01_h128"])
@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_fused_backward_recompute_route_matches_reference_gradients(tag, method, monkeypatch):
    """The same with PSNODE_SAVE_ACTIVATIONS = 0: the backward kernels recompute the forward (K4 / K7 / K9 / K4f / K7f `REC = true`) instead
    of reading what the training forward saved (the default route of these models, covered by the test above)."""
    from py_psnode_amd import autograd as pag
    monkeypatch.setattr(pag, "SAVE_ACTIVATIONS", "0")
    d, m, res, leaves = _run_model(tag, method, "cuda", "require")
    _check(tag, method, d, m, res, leaves, TOL_GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ode02_h64", "dae02_h64", "dae02_z0"])
def test_latent_models_save_their_activations_by_default(tag):
    """hidden-64 direct_encode models: the training forward (K3c) saves, K9 reads; the hidden-16 ones recompute."""
    from py_psnode_amd import autograd as pag
    from py_psnode_amd import fused
    seen = []
    orig_o, orig_d = fused.ode_backward, fused.dae_backward
    try:
        fused.ode_backward = lambda *a, **k: (seen.append(k.get("saved") is not None), orig_o(*a, **k))[1]
        fused.dae_backward = lambda *a, **k: (seen.append(k.get("saved") is not None), orig_d(*a, **k))[1]
        _run_model(tag, "rk4", "cuda", "require")
    finally:
        fused.ode_backward, fused.dae_backward = orig_o, orig_d
    assert seen == [tag.endswith("_h64")], seen


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,tag", [("generic", "ode01"), ("mfma", "ode01"), ("wide", "ode01"), ("wide", "ode01_h128"), ("wide", "ode01_h32"),
                                        ("generic", "ode01_h128"),
                                        ("saved", "ode01"), ("saved", "ode01_h128"), ("saved", "ode01_h32")])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_ode_backward_kernels_raw_api_vs_reference_gradients(method, kernel, tag):
    """K5 (generic) and K4f (mfma / wide: the one-launch backward at hidden 32 / 64 / 128, recompute and saved forms) through
    the raw-tensor API against the reference's ODE_01 gradients at the matching --hidden."""
    from py_psnode_amd import fused
    d = load(f"g7_grad_{tag}.npz")
    de = [(w.cuda(), b.cuda()) for w, b in layers(d, "sd__de_func__x_dot")]
    t, x, z = tm(d["t"]).cuda(), tm(d["x"]).cuda(), tm(d["z"]).cuda()
    ev, zj = T(d["event_t"]).cuda(), T(d["z_jump"]).cuda()
    a0 = torch.cat((x[0], z[0]), -1)
    xs = fused.ode_integrate(method, de, t, x, z, a0, event_t=ev, z_jump=zj)
    _close(xs.permute(1, 0, 2), d[f"{method}_out0"], "xs", TOL_GPU)
    G = tm(d["G0"]).contiguous().cuda()
    tab = fused.event_table(t, ev)
    saved = None
    if kernel == "saved":       # K4f fed with the activations the forward saved (no recompute)
        xs2, saved = fused.ode_integrate(method, de, t, x, z, a0, event_t=ev, z_jump=zj, save=True)
        _close(xs2, xs.cpu(), "training forward (saving) vs inference forward", TOL_GPU)      # (plain vs log2e-scaled ELU domain: roundings differ)
    gx0, gz, gzj, ga0, gp = fused.ode_backward(method, de, t, z, a0, xs, G, event_idx=tab, z_jump=zj,
                                               kernel="auto" if kernel == "saved" else kernel, saved=saved)
    gx_ref = T(d[f"{method}_g_x"])                       # [B,T,xd]: only x[:,0] carries gradient (directly + via all_initial)
    _close(gx0 + ga0[:, :8], gx_ref[:, 0], "grad x0", TOL_GPU)
    gz_tot = gz.clone()
    gz_tot[0] += ga0[:, 8:]
    _close(gz_tot.permute(1, 0, 2), d[f"{method}_g_z"], "grad z", TOL_GPU)
    _close(gzj, d[f"{method}_g_z_jump"], "grad z_jump", TOL_GPU)
    names = [f"{method}_gp__de_func__x_dot__{2 * k}__{w}" for k in range(4) for w in ("weight", "bias")]
    for a, n in zip(gp, names):
        _close(a, d[n], n, TOL_GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["generic", "mfma"])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_dae_backward_kernels_raw_api_vs_reference_gradients(method, kernel):
    """K5 (generic) and K7 (mfma) through the raw-tensor API against the reference's DAE_01 parameter gradients
    (DE and AE; Init_Func's follow from grad x_init / all_initial and are covered by the model-level test)."""
    from py_psnode_amd import fused
    d = load("g7_grad_dae01.npz")
    de = [(w.cuda(), b.cuda()) for w, b in layers(d, "sd__de_func__x_dot")]
    ae = [(w.cuda(), b.cuda()) for w, b in layers(d, "sd__ae_func__i_calculator")]
    init = layers(d, "sd__init_func__init_fun")
    t, x, z, v, i = (tm(d[k]).cuda() for k in ("t", "x", "z", "v", "i"))
    ev, zj, vj = T(d["event_t"]).cuda(), T(d["z_jump"]).cuda(), T(d["v_jump"]).cuda()
    u = torch.cat((z[0], v[0], i[0]), -1).cpu()
    for k, (w, b) in enumerate(init):
        u = torch.nn.functional.linear(u, w, b)
        if k + 1 < len(init):
            u = torch.nn.functional.elu(u)
    x_init = u.cuda()
    a0 = torch.cat((x_init, z[0], v[0], i[0]), -1)
    xs, is_ = fused.dae_integrate(method, de, ae, x_init, t, x, z, v, i, a0, event_t=ev, z_jump=zj, v_jump=vj)
    _close(xs.permute(1, 0, 2), d[f"{method}_out0"], "xs", TOL_GPU)
    _close(is_.permute(1, 0, 2), d[f"{method}_out1"], "is", TOL_GPU)
    Gx, Gi = tm(d["G0"]).contiguous().cuda(), tm(d["G1"]).contiguous().cuda()
    tab = fused.event_table(t, ev)
    gr = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel=kernel)
    for grp, pre in (("de", "de_func__x_dot"), ("ae", "ae_func__i_calculator")):
        names = [f"{method}_gp__{pre}__{2 * k}__{w}" for k in range(4) for w in ("weight", "bias")]
        for a, n in zip(gr[grp], names):
            _close(a, d[n], n, TOL_GPU)
    _close(gr["z_jump"], d[f"{method}_g_z_jump"], "grad z_jump", TOL_GPU)
    _close(gr["v_jump"], d[f"{method}_g_v_jump"], "grad v_jump", TOL_GPU)
