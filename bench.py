#!/usr/bin/env python3
"""bench.py -- integrated state-steps/sec of the fused HIP integrator (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

N>1 runs one rank per GPU over RCCL.  Either launcher works: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / WORLD_SIZE in the environment), or plain `python bench.py --gpus N`, which re-launches itself under
torch.distributed.run (127.0.0.1 rendezvous, a free port) and passes the ranks' output through -- rank 0 prints the JSON line.

One "step" = one pass of the hot path over one batch: a full integrate_ODE of B trajectories over T-1 grid
steps (default workload = BASELINE.json configs[1]: ODE_01 RK4, B=4096, T=1001, x8 z2 H64, fp32), inputs
already resident in HBM.  At N>1 every rank integrates its own B trajectories (weak scaling, no data-path
collective inside the integration) and an RCCL all-gather reassembles the [T, N*B, xd] batch "for the loss"
(north_star); the timed region covers both.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel vs the fp32 MFMA/VALU peak (157.3 TFLOP/s) -- the bound that binds this path
                (arithmetic intensity ~1900 flop/B); `hbm_*` fields give the HBM view north_star
                also asks for.  Kernel time is measured live with HIP events around every launch in the timed region.
  cpu_baseline  oracle/psnode_oracle.py (PyTorch-CPU restatement of the reference, "port") timed on this host's
                cores on a bounded sample of the same workload.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3     # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0        # HBM3E spec

WORKLOADS = {
    # name: (kind, B, T, dims, H, hidden layers)
    "ode01": dict(kind="ode", B=4096, T=1001, xd=8, zd=2, H=64, nh=3),
    "dae01": dict(kind="dae", B=4096, T=1001, xd=8, zd=2, vd=2, id=2, H=64, nh=3),
    "ode02_latent16": dict(kind="ode", B=4096, T=1001, xd=16, zd=16, H=16, nh=1),
    "ode02_latent64": dict(kind="ode", B=4096, T=1001, xd=64, zd=64, H=64, nh=1),   # hidden_dim 64 (generic kernel)
    # BASELINE config 3: whole ODE_02 direct_encode forward (enc x, enc z, latent integrate, dec pred, dec recon), H=16
    "ode02": dict(kind="ode02_model", B=4096, T=1001, xd=8, zd=2, H=16, nh=1),
    # BASELINE's literal "enc/dec 64 -> 16 latent": encoders / decoder of hidden 64 on the MFMA row kernels (K3b) around the 16-wide
    # latent integrator (K3f); an extension kwarg of models.ODE_Model -- upstream has ONE hidden_dim for all three
    "ode02_enc64": dict(kind="ode02_model", B=4096, T=1001, xd=8, zd=2, H=16, E=64, nh=1),
    # the whole DAE_02 direct_encode forward as the script ships it (hidden 64, neural_01_DAE_02_direct_encode.py:267):
    # Init_Func, four encoders, latent integrate_DAE, both decoders of the solution, both reconstructions -- one launch behind Init_Func (K3g)
    "dae02": dict(kind="dae02_model", B=4096, T=1001, xd=8, zd=2, vd=2, id=2, H=64, nh=1),
    # round 6: data-defined dims outside the specialised integrators' classes (x_dim > 16; z + v + i > 8) -> the generic integrator K0,
    # on MFMA since round 6 (neural_00_ODE_01_no_encode.py:293: the dims come from the npz)
    "ode01_x20": dict(kind="ode", B=4096, T=1001, xd=20, zd=2, H=64, nh=3),
    "dae01_zvi16": dict(kind="dae", B=4096, T=1001, xd=8, zd=4, vd=6, id=6, H=64, nh=3),
}


def mlp(dims, gen_seed):
    torch.manual_seed(gen_seed)
    lin = [torch.nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]
    return [(l.weight.detach().clone(), l.bias.detach().clone()) for l in lin]


def mlp_macs(layers):
    return sum(w.shape[0] * w.shape[1] for w, _ in layers)


def make_problem(w, B, T, seed_offset=0):
    """Synthetic batch in the scripts' layout: B-major [B,T,D] memory, solver gets permute(1,0,2) views."""
    g = torch.Generator().manual_seed(1 + seed_offset)
    xd, zd = w["xd"], w["zd"]
    vd, idim = w.get("vd", 0), w.get("id", 0)
    n = xd + zd + vd + idim
    if w["kind"] == "ode02_model":
        from py_psnode_amd import models
        from py_psnode_amd import neural_dae as nd
        torch.manual_seed(0)
        model = models.ODE_Model(xd, zd, w["H"], direct_encode=True, solver=nd.RK4(), enc_hidden=w.get("E"))
        model.solver.fused = "require"
        p = dict(model=model, de=[(l.weight.detach(), l.bias.detach()) for l in model.de_func.x_dot if isinstance(l, torch.nn.Linear)])
    elif w["kind"] == "dae02_model":
        from py_psnode_amd import models
        from py_psnode_amd import neural_dae as nd
        torch.manual_seed(0)
        model = models.DAE_Model(xd, zd, vd, idim, w["H"], direct_encode=True, solver=nd.RK4())
        model.solver.fused = "require"
        # the library's default route (row kernels + K3c); PSNODE_DAE02_ONE_LAUNCH=1 times the opt-in one-launch K3g (LATE_EXTRAS labels both)
        model.one_launch = None if "PSNODE_DAE02_ONE_LAUNCH" not in os.environ else os.environ["PSNODE_DAE02_ONE_LAUNCH"] == "1"
        lin = lambda seq: [(l.weight.detach(), l.bias.detach()) for l in seq if isinstance(l, torch.nn.Linear)]
        p = dict(model=model, de=lin(model.de_func.x_dot), ae=lin(model.ae_func.i_calculator))
    else:
        p = dict(de=mlp([3 * n] + [w["H"]] * w["nh"] + [xd], 0))
    p["t"] = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1)
    p["x"] = torch.zeros(B, T, xd)
    p["x"][:, 0] = 0.1 * torch.randn(B, xd, generator=g)
    p["z"] = 0.1 * torch.randn(B, T, zd, generator=g)
    p["event_t"] = torch.full((B, 2, 1), -1.0)          # "no events" as the scripts encode it
    p["z_jump"] = torch.zeros(B, 2, zd)
    if w["kind"] == "ode02_model":
        p["a0"] = torch.zeros(1)
    elif w["kind"] == "dae02_model":
        p["a0"] = torch.zeros(1)
        p["x"] = 0.1 * torch.randn(B, T, xd, generator=g)           # the reconstruction reads every row
        p["v"] = 0.1 * torch.randn(B, T, vd, generator=g)
        p["i"] = 0.1 * torch.randn(B, T, idim, generator=g)
        p["v_jump"] = torch.zeros(B, 2, vd)
    elif w["kind"] == "dae":
        p["ae"] = mlp([n + xd + zd + vd] + [w["H"]] * w["nh"] + [idim], 7)
        p["v"] = 0.1 * torch.randn(B, T, vd, generator=g)
        p["i"] = 0.1 * torch.randn(B, T, idim, generator=g)
        p["v_jump"] = torch.zeros(B, 2, vd)
        p["x_init"] = p["x"][:, 0].clone()
        p["a0"] = torch.cat((p["x_init"], p["z"][:, 0], p["v"][:, 0], p["i"][:, 0]), -1)
    else:
        p["a0"] = torch.cat((p["x"][:, 0], p["z"][:, 0]), -1)
    return p


def to_dev(p, dev):
    import copy
    out = {}
    for k, v in p.items():
        if isinstance(v, torch.nn.Module):
            out[k] = copy.deepcopy(v).to(dev)
        else:
            out[k] = [(a.to(dev), b.to(dev)) for a, b in v] if isinstance(v, list) else v.to(dev)
    return out


def tmv(a):
    return a.permute(1, 0, 2)


def run_fused(fused, w, p, method, kernel):
    if w["kind"] == "ode02_model":
        with torch.no_grad():
            return p["model"](t=p["t"], x=p["x"], z=p["z"], event_t=p["event_t"], z_jump=p["z_jump"])[:1]
    if w["kind"] == "dae02_model":
        with torch.no_grad():
            return p["model"](t=p["t"], x=p["x"], z=p["z"], v=p["v"], i=p["i"], event_t=p["event_t"], z_jump=p["z_jump"], v_jump=p["v_jump"])
    if w["kind"] == "ode":
        return (fused.ode_integrate(method, p["de"], tmv(p["t"]), tmv(p["x"]), tmv(p["z"]), p["a0"],
                                    event_t=p["event_t"], z_jump=p["z_jump"], kernel=kernel),)
    return fused.dae_integrate(method, p["de"], p["ae"], p["x_init"], tmv(p["t"]), tmv(p["x"]), tmv(p["z"]), tmv(p["v"]),
                               tmv(p["i"]), p["a0"], event_t=p["event_t"], z_jump=p["z_jump"], v_jump=p["v_jump"], kernel=kernel)


def run_oracle(O, w, p, method, T):
    sl = lambda a: tmv(a)[:T]
    if w["kind"] == "ode02_model":
        import torch.nn.functional as F
        m = p["model"]
        seq = lambda s, a: F.linear(F.elu(F.linear(a, s[0].weight, s[0].bias)), s[2].weight, s[2].bias)
        with torch.no_grad():
            x, z = p["x"][:, :T], p["z"][:, :T]
            Xh, Zh = tmv(seq(m.x_encoder, x)), tmv(seq(m.z_encoder, z))
            de = [(l.weight, l.bias) for l in m.de_func.x_dot if isinstance(l, torch.nn.Linear)]
            sol = O.integrate_ode(method, de, sl(p["t"]), Xh, Zh, torch.cat((Xh[0], Zh[0]), -1), p["event_t"], seq(m.z_encoder, p["z_jump"]))
            return seq(m.x_decoder, sol), seq(m.x_decoder, Xh)
    if w["kind"] == "dae02_model":
        import torch.nn.functional as F
        m = p["model"]
        seq = lambda s, a: F.linear(F.elu(F.linear(a, s[0].weight, s[0].bias)), s[2].weight, s[2].bias)
        lin = lambda s: [(l.weight, l.bias) for l in s if isinstance(l, torch.nn.Linear)]
        with torch.no_grad():
            x, z, v, i = (p[k][:, :T] for k in "xzvi")
            x0 = m.init_func(z0=z[:, 0], v0=v[:, 0], i0=i[:, 0])
            Xh0, Xh, Zh, Vh, Ih = seq(m.x_encoder, x0), tmv(seq(m.x_encoder, x)), tmv(seq(m.z_encoder, z)), tmv(seq(m.v_encoder, v)), tmv(seq(m.i_encoder, i))
            Xs, Is = O.integrate_dae(method, lin(m.de_func.x_dot), lin(m.ae_func.i_calculator), Xh0, sl(p["t"]), Xh, Zh, Vh, Ih,
                                     torch.cat((Xh0, Zh[0], Vh[0], Ih[0]), -1), p["event_t"], seq(m.z_encoder, p["z_jump"]),
                                     seq(m.v_encoder, p["v_jump"]))
            xp = seq(m.x_decoder, Xs)
            xp[0] = x0
            return xp, seq(m.i_decoder, Is), seq(m.x_decoder, Xh), seq(m.i_decoder, Ih)
    if w["kind"] == "ode":
        return O.integrate_ode(method, p["de"], sl(p["t"]), sl(p["x"]), sl(p["z"]), p["a0"], p["event_t"], p["z_jump"])
    return O.integrate_dae(method, p["de"], p["ae"], p["x_init"], sl(p["t"]), sl(p["x"]), sl(p["z"]), sl(p["v"]), sl(p["i"]),
                           p["a0"], p["event_t"], p["z_jump"], p["v_jump"])


def flops_per_state_step(w, p, method):
    stages = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    if w["kind"] == "dae02_model":
        # EXECUTED flops: with z | v | i frozen over a step (zero-order hold) and the a0 group constant over the batch, only the x block of
        # the latent DE's first layer runs per stage -- the dense count of SURVEY 8(d) (4 x 13 H^2 + 8 H^2 MACs per RK4 step) is 4x the
        # arithmetic any implementation of this model has to do, and a fraction of peak priced on it exceeds 1
        H, xd, zd, vd, idim = w["H"], w["xd"], w["zd"], w["vd"], w["id"]
        nblk = 4 if zd else 3
        lat = stages * 2 * H * H + (nblk - 1) * H * H + nblk * H * H             # DE stages, DE per-step constant, AE head (nblk-1 blocks + L2)
        encdec = sum(d * H + H * H for d in (xd, zd, vd, idim) if d) + 2 * (H * H + H * xd) + 2 * (H * H + H * idim)
        return 2 * (lat + encdec)
    f = 2 * stages * mlp_macs(p["de"])
    if w["kind"] == "ode02_model":   # + enc x, enc z, 2x dec per grid point (2 880 flop at H=16)
        H, xd, zd, E = w["H"], w["xd"], w["zd"], w.get("E", w["H"])
        f += 2 * ((xd * E + E * H) + (zd * E + E * H) + 2 * (H * E + E * xd))
    if w["kind"] == "dae":
        f += 2 * mlp_macs(p["ae"])
    return f


def bytes_per_state_step(w):
    """Compulsory HBM traffic per state-step: read t + external inputs, write the outputs."""
    if w["kind"] == "ode02_model":   # fully fused ideal: read x, z, t; write x_pred, x_re
        return 4 * (w["xd"] + w["zd"] + 1 + 2 * w["xd"])
    if w["kind"] == "dae02_model":   # fully fused: read t, x, z, v, i; write x_pred, i_pred, x_re, i_re
        return 4 * (1 + w["xd"] + w["zd"] + w["vd"] + w["id"] + 2 * w["xd"] + 2 * w["id"])
    rd = 4 * (1 + w["zd"] + w.get("vd", 0))
    wr = 4 * (w["xd"] + w.get("id", 0))
    return rd + wr


def executed_flops_per_state_step(w, p, method, kname):
    """Flops the dispatched kernel ISSUES per state-step (SQ_INSTS_MFMA x flop per instruction / trajectories per wave for the MFMA
    integrators -- the per-step instruction counts are those of profiles/r05g_k1x_rk4_pmc_sq.txt, r05_dae01_pmc_sq.txt; the folded
    VALU count for K3f) next to SURVEY 8(d)'s DENSE count of flops_per_state_step(): L1's `a0 | s - a0 | s` image is folded to one
    block and the external-input block runs once per step, so a kernel executes fewer flops than the dense graph has.  None where no
    per-kernel constant is known (frac_executed is then null)."""
    stages = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    if w["kind"] == "dae02_model":
        return flops_per_state_step(w, p, method)            # that count is already the executed one (see there)
    if w["kind"] == "ode02_model" and kname == "valu_dpp" and w.get("E", w["H"]) == w["H"]:
        H, xd, zd = w["H"], w["xd"], w["zd"]
        lat = stages * 2 * H * H + H * H                      # per stage: x block of L1 + L2; per step: the folded z block
        encdec = (xd * H + H * H) + (zd * H + H * H) + 2 * (H * H + H * xd)
        return 2 * (lat + encdec)
    if kname == "mfma_wave" and w["H"] <= 64 and w["kind"] in ("ode", "dae"):
        # v_mfma_f32_4x4x1_16B_f32 = 16 blocks x 16 MAC = 512 flop per wave of 4 trajectories
        per_stage = 8 + 64 + 64 + 8                            # L1 (x dims), two H->H layers, L4 (split-K)
        if w["kind"] == "ode":
            n_mfma = stages * per_stage + 4 * ((2 * w["zd"] + 3) // 4)
        else:
            n_mfma = stages * per_stage + 16 + 148               # + the DE's per-step constant + the AE head (740 per RK4 step)
        return n_mfma * 512 / 4
    if kname == "generic" and w["kind"] in ("ode", "dae"):
        # K0 issues the dense graph, every layer padded to 16 x 16 tiles of v_mfma_f32_16x16x4_f32
        up = lambda v: (v + 15) // 16 * 16
        pad = lambda ls: 2 * sum(up(wt.shape[0]) * up(wt.shape[1]) for wt, _ in ls)
        return stages * pad(p["de"]) + (pad(p["ae"]) if w["kind"] == "dae" else 0)
    if kname == "mfma" and w["kind"] == "ode" and w["H"] in (32, 64, 128) and w["xd"] <= 8:
        nw = w["H"] // 16                                      # K1: nw waves x 16 trajectories on v_mfma_f32_16x16x4_f32 (2048 flop)
        per_wave_stage = 2 + 2 * 4 * nw + 4
        n_mfma = nw * (stages * per_wave_stage + (2 * w["zd"] + 3) // 4)
        return n_mfma * 2048 / 16
    return None


def roofline_fracs(w, p, method, kname, ss, kernel_ms, train=False):
    """Both flop conventions on every line: frac_dense prices SURVEY 8(d)'s dense count (what `frac` is),
    frac_executed the flops the kernel issues -- pipe utilisation."""
    dense = flops_per_state_step(w, p, method) * (3 if train else 1)
    ex = executed_flops_per_state_step(w, p, method, kname)
    if ex is not None and train:
        ex = None                                               # the backward kernels' issue counts are reported by the train lines themselves
    to_frac = lambda f: f * ss / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS
    return {"frac_dense": to_frac(dense), "frac_executed": to_frac(ex) if ex is not None else None,
            "flop_dense_per_state_step": dense, "flop_executed_per_state_step": ex}


def oracle_subset_check(w, p_cpu, method, outs, nb=32, steps=100):
    """gpu_vs_oracle for one line: the first `nb` trajectories x first `steps` grid steps of the TIMED batch (events are decided by
    trajectory 0, which is among them; trajectories do not interact) through oracle/psnode_oracle.py on the host, against the same slice
    of the fused path's outputs -- under both metrics of tests/helpers.py.  The checker, never the thing measured."""
    from oracle import psnode_oracle as O
    B, T_s = w["B"], min(steps + 1, w["T"])
    nb = min(nb, B)
    sub = {}
    for k, v in p_cpu.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == B:
            sub[k] = v[:nb]
        else:
            sub[k] = v
    old = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    try:
        ref = run_oracle(O, w, sub, method, T_s)
    finally:
        torch.set_num_threads(old)
    ref = ref if isinstance(ref, (tuple, list)) else (ref,)
    model = w["kind"] in ("ode02_model", "dae02_model")
    per_traj, elem = 0.0, 0.0
    for got, r in zip(outs, ref):
        r = r.double()
        if model:                                               # model outputs are [B,T,D]; the oracle's are [T,B,D] or [B,T,D]
            y = got[:nb, :T_s].double().cpu()
            if r.shape != y.shape:
                r = r.permute(1, 0, 2)
            y, r = y.permute(1, 0, 2), r.permute(1, 0, 2)
        else:
            y = got[:T_s, :nb].double().cpu()
        d = (y - r).abs()
        per_traj = max(per_traj, float((d.amax((0, 2)) / r.abs().amax((0, 2)).clamp_min(1e-3)).max()))
        elem = max(elem, float((d / r.abs().clamp_min(1e-3)).max()))
    return {"sample": f"first {nb} trajectories x first {T_s - 1} steps of the timed batch, {len(list(zip(outs, ref)))} output tensor(s)",
            "per_trajectory_rel_err": per_traj, "elementwise_rel_err": elem, "tolerance": "1e-5 per trajectory (north_star)"}


def cpu_baseline(w, p_cpu, method, budget_s=12.0, gpu_out=None):
    """The CPU oracle on the node's host cores (bounded sample).  gpu_out = the fused path's first output [T,B,D] for the same batch:
    its error against the oracle's sample is reported under BOTH metrics of tests/helpers.py (north_star's per-trajectory one and
    SURVEY 8(d)'s elementwise one)."""
    from oracle import psnode_oracle as O
    T_s = min(101, w["T"])               # bounded sample: same batch, first 100 grid steps
    # The path is ~300 small ATen ops per step: all host threads is far from the fastest setting on a many-core
    # node (oversubscription).  Probe a few thread counts on a 10-step sample and give the CPU its best one.
    default_threads = torch.get_num_threads()
    best = (float("inf"), default_threads)
    for nt in sorted({4, 8, 16, 32, 64, default_threads}):
        if nt > (os.cpu_count() or nt):
            continue
        torch.set_num_threads(nt)
        run_oracle(O, w, p_cpu, method, 3)                 # warm-up (thread pool, allocator)
        for _ in range(2):                                  # best of two 10-step probes per thread count
            t0 = time.perf_counter()
            run_oracle(O, w, p_cpu, method, min(T_s, 11))
            best = min(best, (time.perf_counter() - t0, nt))
    n_threads = best[1]
    torch.set_num_threads(n_threads)
    times = []
    t_start = time.perf_counter()
    ref = None
    while len(times) < 9 and (time.perf_counter() - t_start < budget_s or not times):
        t0 = time.perf_counter()
        ref = run_oracle(O, w, p_cpu, method, T_s)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    out = {"value": w["B"] * (T_s - 1) / med, "unit": "state-steps/s", "cores": n_threads, "kind": "port",
           "sample": f"oracle/psnode_oracle.py (PyTorch-CPU fp32, reference op order), same batch B={w['B']}, first {T_s - 1} of "
                     f"{w['T'] - 1} steps, median of {len(times)} runs ({med:.3f} s each), torch threads={n_threads}, "
                     f"host cpus={os.cpu_count()}"}
    if gpu_out is not None and ref is not None:
        r = (ref[0] if isinstance(ref, (tuple, list)) else ref).double()
        y = gpu_out[:r.shape[0]].double().cpu() if gpu_out.shape[0] >= r.shape[0] and gpu_out.shape[1] == r.shape[1] else None
        if y is None and gpu_out.dim() == 3 and gpu_out.shape[0] == r.shape[1]:      # model outputs are [B,T,D]
            y = gpu_out.permute(1, 0, 2)[:r.shape[0]].double().cpu()
        if y is not None and y.shape == r.shape:
            d = (y - r).abs()
            out["gpu_vs_oracle"] = {
                "sample": f"all {r.shape[1]} trajectories x first {r.shape[0] - 1} steps of the timed batch",
                "per_trajectory_rel_err": float((d.amax((0, 2)) / r.abs().amax((0, 2)).clamp_min(1e-3)).max()),
                "elementwise_rel_err": float((d / r.abs().clamp_min(1e-3)).max()),
                "tolerance": "1e-5 per trajectory (north_star); the elementwise figure is SURVEY 8(d)'s metric, under which the "
                             "reference's own fp32 result is 1.5e-5..7e-5 from fp64 (DESIGN.md 'Accuracy gate')"}
    return out


def kernel_name_for(lib, _lib, fused, w, p, method, kernel, dev):
    """Which kernel family `auto` resolves to for this workload (what `config.kernel` and the pmc_traffic.json key say)."""
    B, T = w["B"], w["T"]
    if w["kind"] in ("ode02_model", "dae02_model"):
        auto_kernel = 2
    elif w["kind"] == "ode":
        a = _lib.OdeArgsF32()
        a.method, a.x_dim, a.z_dim, a.T, a.B = fused.METHOD_ID[method], w["xd"], w["zd"], T, B
        a.de = fused._mlp(p["de"], dev, "de", [])
        auto_kernel = lib.psnode_ode_kernel_for(a)
    else:
        a = _lib.DaeArgsF32()
        a.method, a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = fused.METHOD_ID[method], w["xd"], w["zd"], w["vd"], w["id"], T, B
        a.de = fused._mlp(p["de"], dev, "de", [])
        a.ae = fused._mlp(p["ae"], dev, "ae", [])
        auto_kernel = lib.psnode_dae_kernel_for(a)
    kname = kernel if kernel != "auto" else {1: "generic", 2: "mfma", 5: "mfma_wave"}[auto_kernel]      # mfma_wave: K1x
    kname = {"wave": "mfma_wave", "tile": "mfma"}.get(kname, kname)
    if w["kind"] == "dae02_model":    # the two routes of the DAE_02 model forward (py_psnode_amd/models.py: DAE_Model._forward_encoded)
        kname = "k3g" if getattr(p.get("model"), "one_launch", False) else "rows+k3c"
    if kname == "mfma" and w["H"] == 16 and w["kind"] in ("ode", "ode02_model"):
        kname = "valu_dpp"       # K3f: the hidden-16 latent ODE runs on VALU + DPP row broadcasts (psnode_latent_dpp.hip)
    return kname


def traffic_for(workload, method, kname, B, T, H):
    """HBM bytes per launch from the PMC counters of a separate rocprofv3 run of the same command (profiles/pmc_traffic.json:
    FETCH_SIZE / WRITE_SIZE passes, gfx950 correction) -- counters cannot be read from inside this process; null when that
    configuration was not profiled."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None
    try:
        return json.load(open(tpath)).get(f"{workload}:{method}:{kname}:B{B}:T{T}" + (f":H{H}" if H is not None else ""))
    except Exception:
        return None


# The other single-GPU configurations of BASELINE.json (configs[2], configs[3]) and the solver the four scripts ship with
# (Euler: neural_00_ODE_01_no_encode.py:75, neural_01_DAE_01_no_encode.py:92) timed in the default run, after the headline.
EXTRAS = [("dae01", "rk4"), ("ode02", "rk4"), ("ode01", "euler"), ("dae01", "euler")]
# Round 4, behind the training lines: the DAE_02 model forward as shipped (hidden 64) on its default route (row kernels + K3c) and in one
# launch (K3g, opt-in), and the widest --hidden the MFMA integrator carries (256: H->H weights streamed from L2)
LATE_EXTRAS = [("dae02", "rk4", None, {"PSNODE_DAE02_ONE_LAUNCH": "0"}, "row kernels + K3c (default route)"),
               ("dae02", "rk4", None, {"PSNODE_DAE02_ONE_LAUNCH": "1"}, "one launch (K3g, opt-in)"),
               ("ode01", "rk4", 256, {}, "hidden 256: 16 waves per tile, streamed weights")]


def extra_line(lib, _lib, fused, workload, method, dev, steps=10, warmup=10, hidden=None, env=None, note=None):
    """One more workload on the driver's clock: `steps` passes at B=4096 x 1000 steps, every launch bracketed by HIP events on the
    launch stream; the passes as a whole by synchronize + perf_counter.  Same accounting as the headline line."""
    w = dict(WORKLOADS[workload])
    if hidden:
        w["H"] = hidden
    saved_env = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        return _extra_line(lib, _lib, fused, workload, w, method, dev, steps, warmup, note)
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _extra_line(lib, _lib, fused, workload, w, method, dev, steps, warmup, note):
    B, T = w["B"], w["T"]
    p_cpu = make_problem(w, B, T)
    p = to_dev(p_cpu, dev)
    if w["kind"] in ("ode02_model", "dae02_model"):
        from py_psnode_amd import neural_dae as nd
        p["model"].solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
        p["model"].solver.fused, p["model"].solver.kernel = "require", "auto"
    import gc
    gc.collect()                    # the previous workload's cyclic garbage (autograd contexts, modules) goes NOW, not inside the timed loop:
    torch.cuda.empty_cache()        # a collection that frees gigabytes of device memory is a run of synchronous hipFree calls (a 38 ms pass among 0.9 ms ones)
    for _ in range(warmup):
        outs = run_fused(fused, w, p, method, "auto")
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    gc.disable()
    try:
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            outs = run_fused(fused, w, p, method, "auto")
            ev[k][1].record()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    kern = sorted(a.elapsed_time(b) for a, b in ev)
    avg, med = sum(kern) / len(kern), kern[len(kern) // 2]
    ss = B * (T - 1)
    flops, bts = flops_per_state_step(w, p_cpu, method), bytes_per_state_step(w)
    kname = kernel_name_for(lib, _lib, fused, w, p, method, "auto", dev)
    ach = flops * ss / (avg * 1e-3) / 1e12
    bound = "valu_fp32" if kname == "valu_dpp" else "mfma"     # K3f issues no MFMA: it is priced against the same fp32 datapath peak
    if w["kind"] == "dae02_model":
        # ONE convention on every line: `frac` = the DENSE count of SURVEY 8(d) (4 x 13 H^2 + 8 H^2 MACs per RK4
        # step for the latent integrator + the encoders / decoders); the executed count (zero-order hold folded) is frac_executed
        H = w["H"]
        stages = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
        nblk = 4 if w["zd"] else 3
        dense_lat = stages * (3 * nblk * H * H + H * H) + ((nblk - 1 + nblk) * H * H + H * H)
        dense = 2 * dense_lat + (flops - 2 * (stages * 2 * H * H + (nblk - 1) * H * H + nblk * H * H))
        ach_dense = dense * ss / (avg * 1e-3) / 1e12
    else:
        dense, ach_dense = flops, ach
    fr = roofline_fracs(w, p_cpu, method, kname, ss, avg)
    if w["kind"] == "dae02_model":
        fr["frac_dense"], fr["flop_dense_per_state_step"] = ach_dense / PEAK_FP32_TFLOPS, dense
    try:
        check = oracle_subset_check(w, p_cpu, method, outs)
    except Exception as e:      # the checker must never take the measured line down with it
        check = {"error": f"{type(e).__name__}: {e}"}
    return {"workload": f"{workload} {method}: B={B} x {T - 1} steps, H{w['H']}" + (f" [{note}]" if note else ""), "kernel": kname, "steps": steps, "warmup": warmup,
            "value": ss * steps / elapsed, "unit": "state-steps/s", "ms_per_step": elapsed / steps * 1e3,
            "outputs_finite": bool(torch.isfinite(outs[0]).all()),
            "gpu_vs_oracle": check,
            "roofline": {"bound": bound, "achieved": ach_dense, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach_dense / PEAK_FP32_TFLOPS,
                         "frac_dense": fr["frac_dense"], "frac_executed": fr["frac_executed"],
                         "frac_note": "frac = frac_dense: SURVEY 8(d)'s dense flop count over the measured kernel time (it can exceed 1 where folding "
                                      "L1's a0 | s - a0 | s image and the zero-order hold removes most of the dense graph's work); frac_executed "
                                      "= the flops the kernel issues = pipe utilisation",
                         "flop_executed_per_state_step": fr["flop_executed_per_state_step"],
                         "traffic": traffic_for(workload, method, kname, B, T, w["H"] if w["H"] != WORKLOADS[workload]["H"] else None),
                         "traffic_source": "profiles/pmc_traffic.json",
                         "kernel_ms": avg, "kernel_ms_median": med,
                         "flop_per_state_step": dense, "bytes_per_state_step": bts,
                         "hbm_achieved_GBs": bts * ss / (avg * 1e-3) / 1e9}}


class Trainer:
    """One training step of the fused autograd route on a resident batch: forward (saving activations where the policy of
    py_psnode_amd/autograd.py says so) + the scripts' masked-MSE loss + fused backward (+ the sharded reductions at N > 1).
    neural_00_ODE_01_no_encode.py:350-360, neural_01_DAE_01_no_encode.py:405-424."""

    def __init__(self, w, p, method, kernel, loss, dev, dist=None):
        from py_psnode_amd import autograd as pag
        from py_psnode_amd import loss as ploss
        from py_psnode_amd import sharded
        assert w["kind"] in ("ode", "dae"), "training covers the ode01 / dae01 workloads"
        self.w, self.p, self.method, self.kernel, self.loss, self.dev, self.dist = w, p, method, kernel, loss, dev, dist
        self.pag, self.ploss, self.sharded = pag, ploss, sharded
        B, T = w["B"], w["T"]
        mk = lambda ls: [q.clone().requires_grad_(True) for wb in ls for q in wb]
        pair = lambda ps: [(ps[k], ps[k + 1]) for k in range(0, len(ps), 2)]
        self.params = mk(p["de"]) + (mk(p["ae"]) if w["kind"] == "dae" else [])
        nde = 2 * len(p["de"])
        self.layers, self.ae_layers = pair(self.params[:nde]), pair(self.params[nde:])
        self.G = torch.randn(T, B, w["xd"], device=dev)
        # the scripts' datasets: ODE mask defaults to ones like x (neural_base.py:14-32), DAE mask is [N,T,1]
        gm = torch.Generator().manual_seed(11)
        mshape = (B, T, w["xd"]) if w["kind"] == "ode" else (B, T, 1)
        self.mask = (torch.rand(mshape, generator=gm) > 0.1).float().to(dev)

    def loss_of(self, xs, is_=None):
        """The scripts' loss on the [B,T,D] predictions (neural_00_ODE_01_no_encode.py:353-355, neural_01_DAE_01_no_encode.py:414-419)."""
        w, p, m, ploss, sharded = self.w, self.p, self.mask, self.ploss, self.sharded
        mse = torch.nn.functional.mse_loss
        if self.loss == "weighted-sum":
            return (xs * self.G).sum() + (is_.sum() if is_ is not None else 0.0)
        xp = xs.permute(1, 0, 2)
        if self.dist is not None and self.loss == "mse-fused":
            # data-parallel step: global 1/sum(mask) by a scalar all-reduce, local share of the global loss, no gather
            if is_ is None:
                return sharded.masked_mse_sharded(xp, p["x"], m)[0]
            wx = [10.0 if d == 1 else 1.0 for d in range(w["xd"])]
            return (sharded.masked_mse_sharded(xp, p["x"], m, col_weight=wx, t0_weight=1.0)[0]
                    + sharded.masked_mse_sharded(is_.permute(1, 0, 2), p["i"], m, t0_weight=1.0)[0])
        if is_ is None:
            if self.loss == "mse-fused":
                return ploss.ode_loss(xp, p["x"], m)[0]
            return torch.sum(torch.sum(torch.sum(mse(xp, p["x"], reduction="none") * m, dim=1), dim=0) / torch.sum(m))
        ip = is_.permute(1, 0, 2)
        if self.loss == "mse-fused":
            return ploss.dae_loss(xp, p["x"], ip, p["i"], m)[0]
        x, i = p["x"], p["i"]
        x_loss = (torch.sum(mse(xp, x, reduction="none") * m) + torch.sum(mse(xp[:, :, 1:2], x[:, :, 1:2], reduction="none") * m) * 9) / torch.sum(m)
        i_loss = torch.sum(mse(ip, i, reduction="none") * m) / torch.sum(m)
        return x_loss + i_loss + mse(x[:, 0, :], xp[:, 0, :]) + mse(i[:, 0, :], ip[:, 0, :])

    def step(self, marks=None):
        """marks: three more HIP events recorded behind the forward, the loss and the backward (kernel time per family)."""
        w, p, pag = self.w, self.p, self.pag
        for q in self.params:
            q.grad = None
        if w["kind"] == "dae":
            xs, is_ = pag.fused_dae_integrate(self.method, self.kernel, self.layers, self.ae_layers, p["x_init"], tmv(p["t"]),
                                              tmv(p["z"]), tmv(p["v"]), tmv(p["i"]), p["a0"], p["event_t"], p["z_jump"], p["v_jump"])
            outs = (xs.detach(), is_.detach())
        else:
            xs, is_ = pag.fused_ode_integrate(self.method, self.kernel, self.layers, tmv(p["t"]), tmv(p["x"]), tmv(p["z"]), p["a0"],
                                              p["event_t"], p["z_jump"]), None
            outs = (xs.detach(),)
        if marks:
            marks[0].record()
        loss = self.loss_of(xs, is_)
        if marks:
            marks[1].record()
        loss.backward()
        if marks:
            marks[2].record()
        if self.dist is not None:
            self.sharded.all_reduce_param_grads(self.params)
        return outs


# The training step of the no_encode models on the driver's clock: RK4 and the solver the scripts ship with (Euler) at the
# scripts' hidden 64 constructor default... and the argparse default --hidden 128 (neural_00_ODE_01_no_encode.py:245-246)
TRAIN_EXTRAS = [("ode01", "rk4", 64), ("dae01", "rk4", 64), ("ode01", "euler", 64), ("dae01", "euler", 64), ("ode01", "rk4", 128), ("dae01", "rk4", 128)]


def backward_kernel_name(fused, w, p, method):
    """Which backward kernel AUTO runs for a saved-activation training step of this workload (round 6: K4x, the one-wave-per-4-trajectories
    backward, takes the ODE at hidden 33..64 up to 4608 trajectories per GPU; K4f / K7f otherwise)."""
    mfma_class = (w["xd"] <= 16 and w["zd"] <= 8) if w["kind"] == "ode" else (w["xd"] <= 8 and w["zd"] + w["vd"] + w["id"] <= 8)
    if not mfma_class:
        return "k5"           # the generic backward (register path of the DE since round 6)
    if w["kind"] == "dae":
        return "k7f+k7h"
    return "k4x" if (w["B"] <= 4608 and fused.ode_backward_supported(method, p["de"], w["xd"], w["zd"], "wave")) else "k4f"


def train_extra_line(fused, workload, method, hidden, dev, steps=10, warmup=3):
    """One training workload on the driver's clock: `steps` forward + loss + backward passes at B=4096 x 1000 steps.  roofline.frac on
    the 3x-forward flop convention (forward + data gradients + weight gradients); kernel time per family from HIP events between the
    three parts of the step; saved_bytes = the stage activations the forward kept for the backward (0: the backward recomputes)."""
    w = dict(WORKLOADS[workload])
    w["H"] = hidden
    B, T = w["B"], w["T"]
    p_cpu = make_problem(w, B, T)
    p = to_dev(p_cpu, dev)
    tr = Trainer(w, p, method, "auto", "mse-fused", dev)
    for _ in range(warmup):
        tr.step()
    torch.cuda.synchronize(dev)
    E = lambda: torch.cuda.Event(enable_timing=True)
    ev = [(E(), E(), E(), E()) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        outs = tr.step(ev[k][1:])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    part = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in ev) / steps
    whole = sorted(e[0].elapsed_time(e[3]) for e in ev)
    avg, med = sum(whole) / steps, whole[steps // 2]
    ss = B * (T - 1)
    flops = 3 * flops_per_state_step(w, p_cpu, method)
    ach = flops * ss / (avg * 1e-3) / 1e12
    grads_ok = all(q.grad is not None and bool(torch.isfinite(q.grad).all()) for q in tr.params)
    res = {"workload": f"{workload} {method} TRAIN (forward + masked-MSE loss + fused backward): B={B} x {T - 1} steps, H{hidden}",
           "steps": steps, "warmup": warmup, "value": ss * steps / elapsed, "unit": "state-steps/s", "ms_per_step": elapsed / steps * 1e3,
           "outputs_finite": bool(torch.isfinite(outs[0]).all()), "grads_finite": grads_ok,
           "saved_bytes": int(tr.pag.last_saved_bytes),
           "backward_kernel": backward_kernel_name(fused, w, p, method),
           "host_enqueue_ms": None,
           "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                        "flop_convention": "3 x forward flops per state-step", "flop_per_state_step": flops,
                        "kernel_ms": avg, "kernel_ms_median": med, "kernel_ms_each": [round(e[0].elapsed_time(e[3]), 3) for e in ev],
                        "kernel_ms_by_family": {"forward": part(0, 1), "loss": part(1, 2), "backward": part(2, 3)}}}
    # host-side enqueue cost of one step (no device wait inside the step: the difference to the GPU time is what the host may hide)
    torch.cuda.synchronize(dev)
    th = time.perf_counter()
    tr.step()
    res["host_enqueue_ms"] = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize(dev)
    del tr, p, outs
    # autograd contexts hold their saved tensors in reference cycles: without a collection HERE the cyclic collector frees this workload's
    # gigabytes of saved activations (synchronous hipFree calls) in the middle of the NEXT workload's timed loop (a 42 ms step among 8.4 ms ones)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


MODEL_TRAIN_EXTRAS = [("dae02", "rk4"), ("dae02", "euler"), ("ode02", "rk4")]
# Round 6: the direct_encode models at the scripts' argparse default --hidden 128 (neural_00_ODE_02_direct_encode.py:
# 160-162, neural_01_DAE_02_direct_encode.py:246-248): K3w / K9w with their weight gradients still contracted by library GEMMs -- on the line so
# that the cost is visible.  Euler = the solver the scripts ship with.
MODEL_H128_EXTRAS = [("ode02", "euler", 128), ("dae02", "euler", 128)]
# round 6, last: shapes without a specialisation, on the generic integrator K0 (register form)
GENERIC_EXTRAS = [("ode01_x20", "rk4", "x_dim 20: generic integrator K0"), ("dae01_zvi16", "rk4", "z + v + i = 16: generic integrator K0")]
# ... and a training step on them: K0 forward + K6 loss + the generic backward K5 (Euler: the scripts' solver)
GENERIC_TRAIN_EXTRAS = [("ode01_x20", "euler", 64), ("dae01_zvi16", "euler", 64)]


def safe_line(label, fn):
    """An extra workload must never take the measured headline down with it: a failure becomes an entry that says so."""
    try:
        return fn()
    except Exception as e:      # noqa: BLE001
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return {"workload": label, "error": f"{type(e).__name__}: {e}"[:400]}


def model_train_extra_line(workload, method, dev, steps=8, warmup=3, hidden=None):
    """Training step of a direct_encode MODEL as the script runs it (neural_01_DAE_02_direct_encode.py:359-370 / neural_00_ODE_02_direct_encode.py:267-275):
    encoders -> fused latent integrator -> decoders -> the script's loss (K6) -> backward through all of it (row-MLP backward kernels, K9 / K8f),
    B=4096 x 1000 steps at the hidden width the script ships with.  roofline.frac on the 3x-forward executed-flop convention."""
    from py_psnode_amd import loss as L, models
    from py_psnode_amd import neural_dae as nd
    w = dict(WORKLOADS[workload])
    if hidden:
        w["H"] = hidden
    B, T, H = w["B"], w["T"], w["H"]
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
    x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
    ev, zj, vj = -torch.ones(B, 2, 1, device=dev), torch.zeros(B, 2, 2, device=dev), torch.zeros(B, 2, 2, device=dev)
    mask8, mask1 = torch.ones(B, T, 8, device=dev), torch.ones(B, T, 1, device=dev)
    solver = {"rk4": nd.RK4, "euler": nd.Euler, "midpoint": nd.Midpoint}[method]()
    torch.manual_seed(0)
    if workload == "ode02":
        m = models.ODE_Model(8, 2, H, direct_encode=True, solver=solver).to(dev)
    else:
        m = models.DAE_Model(8, 2, 2, 2, H, direct_encode=True, solver=solver).to(dev)
    m.solver.fused = "require"

    def step():
        m.zero_grad(set_to_none=True)
        if workload == "ode02":
            o = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
            loss = L.ode02_loss(o[0], o[1], x, mask8)[0]
        else:
            o = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
            loss = L.dae02_loss(o[0], o[1], o[2], o[3], x, i, mask1)[0]
        loss.backward()
        return o

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    E = lambda: torch.cuda.Event(enable_timing=True)
    evs = [(E(), E()) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        evs[k][0].record()
        outs = step()
        evs[k][1].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    each = sorted(a.elapsed_time(b) for a, b in evs)
    avg, med = sum(each) / steps, each[steps // 2]
    ss = B * (T - 1)
    p_cpu = make_problem(w, 16, 3)
    flops = 3 * flops_per_state_step(w, p_cpu, method)
    ach = flops * ss / (avg * 1e-3) / 1e12
    grads_ok = all(q.grad is not None and bool(torch.isfinite(q.grad).all()) for q in m.parameters())
    res = {"workload": f"{workload} {method} MODEL TRAIN (encoders + latent integrator + decoders + the script's loss + backward): B={B} x {T - 1} steps, H{H}",
           "steps": steps, "warmup": warmup, "value": ss * steps / elapsed, "unit": "state-steps/s", "ms_per_step": elapsed / steps * 1e3,
           "outputs_finite": bool(torch.isfinite(outs[0]).all()), "grads_finite": grads_ok,
           "roofline": {"bound": "mfma" if H >= 64 else "valu_fp32", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                        "flop_convention": "3 x forward executed flops per state-step", "flop_per_state_step": flops,
                        "kernel_ms": avg, "kernel_ms_median": med, "kernel_ms_each": [round(a.elapsed_time(b), 3) for a, b in evs]}}
    del m, outs
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line under torch.distributed.run on this
    node (one process per GPU, RCCL; rendezvous on 127.0.0.1 at a free port) and return its exit code.  The children see
    WORLD_SIZE in their environment and run main() proper; their stdout / stderr pass through, rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n}: launching {n} rank(s) under torch.distributed.run (127.0.0.1:{port})", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7      # MI355X: 7 xGMI links per GPU, ~153 GB/s each direction (MI355X_MICROARCH.md)


def multi_gpu_report(w, world, B, T, step_ms, integrate_ms, gather_ms, chunks, pipelined, chunk_model, gather_algo=None, by_algo=None,
                     algo_choice=None):
    """What a reader needs to interpret an N > 1 line without a second run: the shard each rank contributes, what the all-gather should
    cost on xGMI (every rank receives (N-1) shards; a direct all-gather spreads them over N-1 of the 7 links, a ring pushes them all through
    one), what the two legs cost alone, and how much of the shorter one the pipeline hid."""
    widths = [w["xd"]] + ([w["id"]] if w["kind"] == "dae" else [])
    shard = sum(T * B * d * 4 for d in widths)
    recv = (world - 1) * shard
    direct = shard / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0
    ring = recv / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0
    rep = {"shard_bytes": shard, "received_bytes_per_rank": recv,
           "predicted_gather_ms": {"direct_one_link_per_peer": direct, "ring_one_link": ring,
                                   "assumption": f"{XGMI_LINKS} xGMI links x {XGMI_LINK_GBS} GB/s per GPU, point to point"},
           "integrate_only_ms": integrate_ms, "gather_only_ms": gather_ms, "step_ms": step_ms, "chunks": chunks, "pipelined": bool(pipelined),
           "chunk_model": chunk_model,
           # rccl = all_gather_into_tensor (RCCL picks ring or direct by message size); direct = sharded.all_gather_direct (N-1 point-to-point
           # pairs per chunk, one shard per xGMI link whatever the tuner thinks); by_algo: --gather-algo both, measured after the timed region
           "gather_algo": gather_algo, "by_algo": by_algo, "algo_choice": algo_choice}
    if gather_ms is not None:
        serial = integrate_ms + gather_ms
        rep["serial_ms"] = serial
        rep["hidden_ms"] = serial - step_ms                      # > 0: the pipeline overlapped that much of the two legs
        rep["hidden_frac_of_shorter_leg"] = (serial - step_ms) / max(min(integrate_ms, gather_ms), 1e-9)
        rep["gather_achieved_GBs_per_rank"] = recv / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else None
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ode01", choices=sorted(WORKLOADS))
    ap.add_argument("--method", default="rk4", choices=["euler", "midpoint", "rk4"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "mfma", "tile", "wave"], help="tile / wave: K1 (4-wave tile) / K1x (one wave per 4 trajectories) for the ODE forward")
    ap.add_argument("--batch", type=int, default=None, help="override trajectories per GPU")
    ap.add_argument("--grid", type=int, default=None, help="override grid points T")
    ap.add_argument("--hidden", type=int, default=None, help="override the MLPs' hidden width (the scripts' --hidden)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="default run: skip the extra workloads (configs 3, 4, Euler) timed after the headline")
    ap.add_argument("--no-train-extras", action="store_true", help="default run: skip the training-step workloads among the extras")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather (integrate-only scaling)")
    ap.add_argument("--train", action="store_true", help="time forward + backward (fused autograd route) instead of the forward alone")
    ap.add_argument("--loss", default="mse-fused", choices=["mse-fused", "mse-torch", "weighted-sum"],
                    help="with --train: the scripts' masked MSE through the fused loss kernel (default), the same expression in "
                         "PyTorch ops, or the plain sum(xs * G) of earlier rounds")
    ap.add_argument("--train-baseline-steps", type=int, default=0,
                    help="with --train: also time the unrolled PyTorch-autograd walk on the GPU for this many grid steps")
    ap.add_argument("--force-dist", action="store_true", help="testing: run the N>1 code path (RCCL group, gather) even at world size 1")
    ap.add_argument("--gather-layout", default="chunks", choices=["chunks", "batch"],
                    help="N>1, pipelined: per-chunk rank-major buffers (zero-copy) or every chunk gathered INTO one [T, N*B, D] tensor")
    ap.add_argument("--gather-algo", default="auto", choices=["auto", "rccl", "direct", "both"],
                    help="N>1: the all-gather as RCCL's own all_gather_into_tensor, or as N-1 point-to-point pairs per chunk (sharded.all_gather_direct: "
                         "every peer's shard on its own xGMI link, independent of RCCL's ring/direct choice); 'auto' (default) = both are timed alone in "
                         "the untimed warm-up and the faster one runs the timed region (every rank picks the same one: max over ranks); 'both' = timed "
                         "region on rccl, then both timed again after it (gather alone and the pipelined step) and printed under multi_gpu")
    ap.add_argument("--collective", default="gather", choices=["gather", "loss-only"],
                    help="N>1: what follows the integration -- the all-gather of the output shards (north_star), or SURVEY 8(e)'s cheaper alternative: the "
                         "sharded masked-MSE loss (three scalar all-reduces, nothing gathered)")
    ap.add_argument("--chunks", default="auto", help="N>1: time chunks of the integrate/all-gather pipeline (1 = no overlap; 'auto' = chosen from "
                    "the measured integrate-only and gather-only times of the warm-up)")
    args = ap.parse_args()
    requested_gpus = args.gpus

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_dist):
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != requested_gpus and not args.force_dist:
        # a line that says n_gpus = 8 must have run on 8 ranks: a launcher / --gpus mismatch is an error, not a silently smaller job
        sys.exit(f"bench.py: --gpus {requested_gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the fused integrator has no CPU path)")
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible) -- one process per GPU")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)

    from py_psnode_amd import _lib, fused
    lib = _lib.load()

    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["B"] = args.batch
    if args.hidden:
        w["H"] = args.hidden
    if args.grid:
        w["T"] = args.grid
    B, T = w["B"], w["T"]
    p_cpu = make_problem(w, B, T, seed_offset=rank)
    p = to_dev(p_cpu, dev)
    if w["kind"] in ("ode02_model", "dae02_model"):
        from py_psnode_amd import neural_dae as nd
        for mdl in (p["model"], p_cpu["model"]):
            mdl.solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[args.method]()
            mdl.solver.fused, mdl.solver.kernel = "require", args.kernel
    n_out = 1 if w["kind"] == "ode" else (4 if w["kind"] == "dae02_model" else 2)
    from py_psnode_amd import sharded
    loss_only = (world > 1 or args.force_dist) and args.collective == "loss-only" and not args.train and w["kind"] in ("ode", "dae")
    do_gather = (world > 1 or args.force_dist) and not args.no_gather and not args.train and not loss_only   # training never gathers (sharded loss)
    algo = "rccl" if args.gather_algo in ("both", "auto") else args.gather_algo
    algo_choice = None
    loss_helper = Trainer(w, p, args.method, args.kernel, "mse-fused", dev, dist) if loss_only else None
    auto_chunks = str(args.chunks) == "auto"
    args.chunks = 4 if auto_chunks else int(args.chunks)
    pipelined = do_gather and w["kind"] in ("ode", "dae") and args.chunks > 1
    gathered = None
    if do_gather and not pipelined:
        widths = [w["xd"]] + ([w["id"]] if w["kind"] == "dae" else [])
        gathered = [torch.empty((world * T, B, d), dtype=torch.float32, device=dev) for d in widths]

    if do_gather and args.gather_layout == "batch" and args.gather_algo == "auto":
        args.gather_algo = "rccl"                # (the list all_gather into strided views is RCCL's own: nothing to choose)
    if do_gather and args.gather_layout == "batch" and args.gather_algo != "rccl":
        sys.exit("bench.py: --gather-layout batch gathers through c10d's list all_gather (RCCL's own algorithm); use --gather-layout chunks with --gather-algo direct / both")
    trainer = Trainer(w, p, args.method, args.kernel, args.loss, dev, dist) if args.train else None

    step_algo = [algo]       # (a cell: the 'both' leg re-times the step on the other algorithm after the timed region)

    def one_step(ev_pair=None):
        """One pass of the hot path (+ the all-gather at N>1).  ev_pair brackets the compute-stream kernels only."""
        if ev_pair:
            ev_pair[0].record()
        if args.train:
            outs = trainer.step()
            if ev_pair:
                ev_pair[1].record()
            return outs
        if pipelined:
            # trajectory 0 of the GLOBAL batch decides the event steps for everybody (neural_base.py:54,61): rank 0 builds the
            # table, every rank receives it -- SURVEY 8(e); a per-rank table would let each shard decide for itself
            tab = sharded.broadcast_event_table(tmv(p["t"]), p["event_t"])
            if w["kind"] == "ode":
                xs, _, works = sharded.integrate_ode_pipelined(args.method, p["de"], tmv(p["t"]), tmv(p["x"]), tmv(p["z"]), p["a0"],
                                                               event_idx=tab, z_jump=p["z_jump"], chunks=args.chunks, wait=False,
                                                               kernel=args.kernel, check_shards=False, layout=args.gather_layout, algo=step_algo[0])
                outs = (xs,)
            else:
                outs, _, works = sharded.integrate_dae_pipelined(args.method, p["de"], p["ae"], p["x_init"], tmv(p["t"]), tmv(p["z"]),
                                                                 tmv(p["v"]), tmv(p["i"]), p["a0"], event_idx=tab, z_jump=p["z_jump"],
                                                                 v_jump=p["v_jump"], chunks=args.chunks, wait=False, kernel=args.kernel,
                                                                 check_shards=False, layout=args.gather_layout, algo=step_algo[0])
            if ev_pair:
                ev_pair[1].record()
            for wk in works:
                wk.wait()
            return outs
        outs = run_fused(fused, w, p, args.method, args.kernel)
        if ev_pair:
            ev_pair[1].record()
        if gathered is not None:
            for g, o in zip(gathered, outs):
                sharded._gather_into(g, o, None, step_algo[0], False)
        if loss_only:      # the sharded loss instead of the gather: sum(mask), then the loss terms -- scalar-sized all-reduces only
            with torch.no_grad():
                loss_helper.loss_of(outs[0], outs[1] if len(outs) > 1 else None)
        return outs

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if do_gather:
        sharded.require_equal_shards(B, dev)     # once, outside the timed region: every rank holds B trajectories (weak scaling)
    if do_gather and args.gather_algo == "auto" and w["kind"] in ("ode", "dae"):
        # --gather-algo auto: time the gather of the full shards alone on both algorithms (untimed region) and keep the faster one.  xGMI is
        # a point-to-point mesh: if RCCL's tuner rings 131 MB messages the direct exchange wins by up to (N - 1) x; if it already spreads the
        # shards over the links the two tie and RCCL's own path stays.  Max over ranks, so that every rank picks the same algorithm.
        outs_a = run_fused(fused, w, p, args.method, args.kernel)
        bufs_a = [torch.empty((world * T, B, o.shape[-1]), dtype=torch.float32, device=dev) for o in outs_a]
        srcs_a = [o.contiguous() for o in outs_a]
        probe = {}
        for al in ("rccl", "direct"):
            for f_, o in zip(bufs_a, srcs_a):
                sharded._gather_into(f_, o, None, al, False)
            fence()
            t_a = time.perf_counter()
            for _ in range(2):
                for f_, o in zip(bufs_a, srcs_a):
                    sharded._gather_into(f_, o, None, al, False)
            fence()
            probe[al] = (time.perf_counter() - t_a) / 2 * 1e3
        tt = torch.tensor([probe["rccl"], probe["direct"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        probe = {"rccl": float(tt[0]), "direct": float(tt[1])}
        algo = "direct" if probe["direct"] < 0.9 * probe["rccl"] else "rccl"       # (a tie keeps RCCL's own collective)
        step_algo[0] = algo
        algo_choice = {"probe_gather_only_ms": probe, "picked": algo, "rule": "direct if it is > 10 % faster than rccl in the warm-up probe"}
        del bufs_a, srcs_a, outs_a
    chunk_model = None
    if do_gather and w["kind"] in ("ode", "dae") and auto_chunks:
        # --chunks auto: measure the two legs alone (untimed region), then pick the chunk count of the pipeline from them.  Model: with c
        # chunks the shorter leg hides behind the longer one except for its first / last chunk, and every chunk costs one more launch +
        # collective start (~0.05 ms): total(c) = max(I, G) + min(I, G) / c + 0.05 c  ->  c* = sqrt(min(I, G) / 0.05), clamped to 1..16
        outs0 = run_fused(fused, w, p, args.method, args.kernel)
        fence()
        t_i = time.perf_counter()
        for _ in range(2):
            outs0 = run_fused(fused, w, p, args.method, args.kernel)
        fence()
        i_ms = (time.perf_counter() - t_i) / 2 * 1e3
        bufs = [torch.empty((world * T, B, o.shape[-1]), dtype=torch.float32, device=dev) for o in outs0]
        for f_, o in zip(bufs, outs0):
            sharded._gather_into(f_, o.contiguous(), None, algo, False)
        fence()
        t_g = time.perf_counter()
        for _ in range(2):
            for f_, o in zip(bufs, outs0):
                sharded._gather_into(f_, o.contiguous(), None, algo, False)
        fence()
        g_ms = (time.perf_counter() - t_g) / 2 * 1e3
        tt = torch.tensor([i_ms, g_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # every rank must pick the SAME count
        i_ms, g_ms = float(tt[0]), float(tt[1])
        c_star = int(round((min(i_ms, g_ms) / 0.05) ** 0.5))
        args.chunks = max(1, min(16, c_star))
        pipelined = args.chunks > 1
        chunk_model = {"integrate_only_ms": i_ms, "gather_only_ms": g_ms, "per_chunk_overhead_ms": 0.05, "chunks": args.chunks,
                       "predicted_total_ms": max(i_ms, g_ms) + min(i_ms, g_ms) / args.chunks + 0.05 * args.chunks}
        if not pipelined:
            widths = [w["xd"]] + ([w["id"]] if w["kind"] == "dae" else [])
            gathered = [torch.empty((world * T, B, d), dtype=torch.float32, device=dev) for d in widths]
        del bufs, outs0
    for _ in range(args.warmup):
        one_step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        outs = one_step(ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    kern_med_ms = kern_ms[len(kern_ms) // 2]
    if dist is not None:
        tt = torch.tensor([elapsed, kern_avg_ms, kern_med_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kern_avg_ms, kern_med_ms = float(tt[0]), float(tt[1]), float(tt[2])

    gather_only_ms, by_algo = None, None
    if do_gather and w["kind"] in ("ode", "dae"):
        # config 5 also wants the gather alone: un-pipelined all-gathers of the full [T,B,D] shards, after the timed region -- on the timed
        # region's algorithm, and with --gather-algo both on the other one too (+ the pipelined step re-timed on it)
        flats = [torch.empty((world * T, B, o.shape[-1]), dtype=torch.float32, device=dev) for o in outs]
        srcs = [o.contiguous() for o in outs]

        def gather_alone(al):
            for f_, o in zip(flats, srcs):
                sharded._gather_into(f_, o, None, al, False)
            fence()
            tg = time.perf_counter()
            for _ in range(3):
                for f_, o in zip(flats, srcs):
                    sharded._gather_into(f_, o, None, al, False)
            fence()
            ms = (time.perf_counter() - tg) / 3 * 1e3
            if dist is not None:
                tt_ = torch.tensor([ms], dtype=torch.float64, device=dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                ms = float(tt_[0])
            return ms

        gather_only_ms = gather_alone(algo)
        if args.gather_algo == "both":
            other = "direct" if algo == "rccl" else "rccl"
            by_algo = {"gather_only_ms": {algo: gather_only_ms, other: gather_alone(other)}, "step_ms": {algo: elapsed / args.steps * 1e3}}
            if args.gather_layout == "chunks" or not pipelined:
                step_algo[0] = other
                for _ in range(max(1, args.warmup)):
                    one_step()
                fence()
                ts = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                fence()
                ms = (time.perf_counter() - ts) / args.steps * 1e3
                tt_ = torch.tensor([ms], dtype=torch.float64, device=dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                by_algo["step_ms"][other] = float(tt_[0])
                step_algo[0] = algo
        del flats, srcs

    finite = bool(torch.isfinite(outs[0]).all())
    state_steps_launch = B * (T - 1)
    flops = flops_per_state_step(w, p_cpu, args.method)
    if args.train:
        flops *= 3   # forward + recomputed forward... counted as the usual fwd + 2x bwd GEMM work (data + weight gradients)
    bts = bytes_per_state_step(w)
    value = world * state_steps_launch * args.steps / elapsed
    ach_tf = flops * state_steps_launch / (kern_avg_ms * 1e-3) / 1e12
    ach_gbs = bts * state_steps_launch / (kern_avg_ms * 1e-3) / 1e9

    if rank == 0:
        kname = kernel_name_for(lib, _lib, fused, w, p, args.method, args.kernel, dev)
        traffic = None if args.train else traffic_for(args.workload, args.method, kname, B, T, args.hidden)
        res = {
            "metric": "integrated state-steps/sec (batch x steps/s), RK4 neural-ODE, batch 4096" if (args.workload, args.method, B, w["H"]) == ("ode01", "rk4", 4096, 64)
                      else f"integrated state-steps/sec, {args.workload} {args.method}, batch {B}, hidden {w['H']}",
            "value": value, "unit": "state-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median_kernel": kern_med_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload} {args.method}: B={B} trajectories/GPU x {T - 1} steps, x{w['xd']} z{w['zd']}"
                                   + (f" v{w['vd']} i{w['id']}" if w["kind"] == "dae" else "") + f" H{w['H']}, fp32, h=0.01, no events",
                       "kernel": kname, "trajectories_total": world * B,
                       "world_size_seen": dist.get_world_size() if dist is not None else 1, "device_count": torch.cuda.device_count(),
                       "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if dist is not None else None,
                       "collective": (((("rccl all_gather" if algo == "rccl" else "direct all-gather (N-1 point-to-point pairs per chunk over RCCL send/recv)")
                                        + f" of the output shards [T,B,D], {args.chunks} time chunks overlapped with the integration")
                                       if pipelined else ("rccl all_gather" if algo == "rccl" else "direct all-gather (N-1 point-to-point pairs)")
                                       + " of the output shards [T,B,D]") if do_gather
                                      else ("rccl all_reduce x3 (sum(mask), loss terms): sharded masked-MSE loss, nothing gathered (SURVEY 8(e) alternative)"
                                            if loss_only else "none")),
                       "gather_algo": algo if do_gather else None,
                       "outputs_finite": finite, "integrate_only_ms": kern_avg_ms, "gather_only_ms": gather_only_ms},
            "multi_gpu": multi_gpu_report(w, world, B, T, elapsed / args.steps * 1e3, kern_avg_ms, gather_only_ms, args.chunks if do_gather else None,
                                          pipelined, chunk_model, algo if do_gather else None, by_algo, algo_choice) if dist is not None else None,
            "roofline": {"bound": "valu_fp32" if kname == "valu_dpp" else "mfma", "achieved": ach_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command)" if traffic is not None else None,
                         **{k: v for k, v in roofline_fracs(w, p_cpu, args.method, kname, state_steps_launch, kern_avg_ms, args.train).items()},
                         "kernel_ms": kern_avg_ms, "kernel_ms_median": kern_med_ms,
                         "kernel_ms_min": kern_ms[0], "kernel_ms_max": kern_ms[-1], "flop_per_state_step": flops,
                         "hbm_achieved_GBs": ach_gbs, "hbm_peak_GBs": PEAK_HBM_GBS, "hbm_frac": ach_gbs / PEAK_HBM_GBS,
                         "bytes_per_state_step": bts},
        }
        if args.train:
            res["metric"] = f"training state-steps/sec (forward + backward), {args.workload} {args.method}, batch {B}"
            res["config"]["workload"] += f" | forward + loss ({args.loss}) + fused backward"
            if dist is not None:
                res["config"]["collective"] = "rccl all_reduce: sum(mask) (1 float), loss terms, flat parameter-gradient bucket; no gather"
            if args.train_baseline_steps > 0 and w["kind"] == "ode":
                # the route the reference's scripts take: unrolled autograd through the per-step Python loop, on this GPU
                from py_psnode_amd import models
                from py_psnode_amd import neural_dae as nd
                Ts = args.train_baseline_steps + 1
                de = models.DE_Func(w["xd"] + w["zd"], (w["H"],) * w["nh"], w["xd"]).to(dev)
                solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[args.method]()
                solver.fused = "off"
                def walk():
                    de.zero_grad()
                    xs = solver.integrate_ODE(x_func=de, t=tmv(p["t"])[:Ts], x=tmv(p["x"])[:Ts], z=tmv(p["z"])[:Ts], all_initial=p["a0"])
                    (xs * trainer.G[:Ts]).sum().backward()
                walk(); torch.cuda.synchronize(dev)
                t0 = time.perf_counter(); walk(); torch.cuda.synchronize(dev)
                dtw = time.perf_counter() - t0
                res["autograd_walk_gpu"] = {"value": B * (Ts - 1) / dtw, "unit": "state-steps/s", "sample": f"{Ts - 1} steps, unrolled PyTorch autograd on the same GPU",
                                            "fused_over_walk": value / (B * (Ts - 1) / dtw)}
        headline = (args.workload, args.method, B, T, w["H"], args.kernel) == ("ode01", "rk4", 4096, 1001, 64, "auto")
        if world == 1 and dist is None and headline and not args.train and not args.no_extras:
            out0 = outs[0]
            del outs
            res["extra"] = [extra_line(lib, _lib, fused, wl, m, dev) for wl, m in EXTRAS]
            if not args.no_train_extras:
                res["extra"] += [train_extra_line(fused, wl, m, h, dev) for wl, m, h in TRAIN_EXTRAS]
                res["extra"] += [model_train_extra_line(wl, m, dev) for wl, m in MODEL_TRAIN_EXTRAS]
            res["extra"] += [extra_line(lib, _lib, fused, wl, m, dev, steps=5, warmup=3, hidden=h, env=e, note=n) for wl, m, h, e, n in LATE_EXTRAS]
            if not args.no_train_extras:
                for wl, m, h in MODEL_H128_EXTRAS:
                    res["extra"].append(safe_line(f"{wl} {m}: H{h} forward", lambda: extra_line(lib, _lib, fused, wl, m, dev, steps=3, warmup=2, hidden=h,
                                                                                              note="the scripts' argparse default --hidden 128")))
                    res["extra"].append(safe_line(f"{wl} {m} MODEL TRAIN H{h}", lambda: model_train_extra_line(wl, m, dev, steps=3, warmup=2, hidden=h)))
            for wl, m, n in GENERIC_EXTRAS:
                res["extra"].append(safe_line(f"{wl} {m}", lambda: extra_line(lib, _lib, fused, wl, m, dev, steps=3, warmup=2, note=n)))
            if not args.no_train_extras:
                for wl, m, h in GENERIC_TRAIN_EXTRAS:
                    res["extra"].append(safe_line(f"{wl} {m} TRAIN", lambda: train_extra_line(fused, wl, m, h, dev, steps=3, warmup=2)))
            outs = (out0,)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w, p_cpu, args.method, gpu_out=None if args.train else outs[0])
            res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
            if "gpu_vs_oracle" in res["cpu_baseline"]:
                acc = res["cpu_baseline"]["gpu_vs_oracle"]
                res["config"]["rel_err_vs_oracle"] = {"per_trajectory": acc["per_trajectory_rel_err"], "elementwise": acc["elementwise_rel_err"],
                                                      "sample": acc["sample"]}
        if dist is not None and world > 1 and not res["config"]["rccl_version"]:
            sys.exit("bench.py: N > 1 without an RCCL version -- the collective did not run on RCCL")
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the last line on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
