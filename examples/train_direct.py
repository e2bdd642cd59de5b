#!/usr/bin/env python3
"""The scripts' training loop on the MI355X-native path, end to end:

    npz dataset -> ResidentDataset/ResidentLoader (batches formed on the device)
                -> ODE_Model / DAE_Model (fused integrator; direct_encode variants with the row kernels)
                -> the script's loss on the fused loss kernel -> fused backward kernels -> Adam
                -> evaluation (the scripts' per-dimension masked MSE) -> TorchScript export (save_model)

It is the body of neural_00_ODE_01_no_encode.py:339-400 / neural_01_DAE_01_no_encode.py:405-470 without the logging,
plotting and replay-buffer bookkeeping.  With --synthetic it writes a small dataset in the scripts' npz format first
(SURVEY.md App. C), so it runs anywhere a GPU and the built library are present:

    python examples/train_direct.py --model ode01 --synthetic --epochs 3
    python examples/train_direct.py --model dae02 --train-data training.npz --test-data testing.npz --solver euler
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from py_psnode_amd import datapath, loss as L, models  # noqa: E402
from py_psnode_amd import neural_dae as nd  # noqa: E402
from py_psnode_amd.neural_dae.neural_base import DAE_Curves_Sample, ODE_Curves_Sample  # noqa: E402

HIDDEN = {"ode01": 64, "ode02": 16, "dae01": 64, "dae02": 64}      # what the scripts ship with (their debug override)


def write_synthetic(path, n, T, dae, seed):
    """Damped oscillators driven by a piecewise-constant input: enough structure for the loss to go down."""
    r = np.random.default_rng(seed)
    t = np.tile((np.arange(T, dtype=np.float32) * 0.01).reshape(1, T, 1), (n, 1, 1))
    z = np.repeat(r.standard_normal((n, 1, 2)).astype(np.float32) * 0.3, T, axis=1)
    z[:, T // 2:] += 0.2
    w = 2.0 + r.random((n, 1, 8)).astype(np.float32) * 3.0
    ph = r.random((n, 1, 8)).astype(np.float32) * 6.28
    x = (0.3 * np.exp(-0.5 * t) * np.sin(w * t + ph) + 0.1 * z.sum(-1, keepdims=True)).astype(np.float32)
    d = dict(name=np.array([[f"x{k}", "pu"] for k in range(8)], dtype=object), t=t, x=x, z=z,
             event_t=np.full((n, 1, 1), 0.01 * (T // 2), dtype=np.float32), z_jump=z[:, T // 2:T // 2 + 1].copy())
    if dae:
        v = (0.5 * z + 0.05 * r.standard_normal((n, T, 2))).astype(np.float32)
        d.update(v=v, i=(x[:, :, :2] * 0.5 + v * 0.2).astype(np.float32), v_jump=v[:, T // 2:T // 2 + 1].copy(),
                 mask=np.ones((n, T, 1), dtype=np.float32))
    np.savez(path, **d)


def build(model, solver):
    s = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[solver]()
    H = HIDDEN[model]
    if model.startswith("ode"):
        return models.ODE_Model(8, 2, H, direct_encode=model.endswith("02"), solver=s)
    return models.DAE_Model(8, 2, 2, 2, H, direct_encode=model.endswith("02"), solver=s)


def step_loss(model_name, model, batch):
    """forward + THAT script's training loss -- the four differ (py_psnode_amd/loss.py): ODE_01 masked term only
    (neural_00_ODE_01_no_encode.py:353-355), ODE_02 + x0 term + reconstruction (neural_00_ODE_02_direct_encode.py:267-270),
    DAE_01 with the 9x extra weight on x column 1 (neural_01_DAE_01_no_encode.py:414-419), DAE_02 without it and with both
    reconstruction terms (neural_01_DAE_02_direct_encode.py:359-365)."""
    if model_name.startswith("ode"):
        t, x, z, event_t, z_jump, mask = batch
        out = model(t=t, x=x, z=z, event_t=event_t, z_jump=z_jump)
        if model_name == "ode01":
            return L.ode01_loss(out, x, mask)[0]
        return L.ode02_loss(out[0], out[1], x, mask)[0]
    t, x, z, v, i, event_t, z_jump, v_jump, mask = batch
    out = model(t=t, x=x, z=z, v=v, i=i, event_t=event_t, z_jump=z_jump, v_jump=v_jump)
    if model_name == "dae01":
        return L.dae01_loss(out[0], x, out[1], i, mask)[0]
    return L.dae02_loss(out[0], out[1], out[2], out[3], x, i, mask)[0]


@torch.no_grad()
def evaluate(model_name, model, loader):
    """x_loss_total of evalute_model (neural_00_ODE_01_no_encode.py:104-129): masked squared error over sum(mask)"""
    num = torch.zeros((), device=loader.dataset.device)
    den = torch.zeros((), device=loader.dataset.device)
    for batch in loader:
        x, mask = batch[1], batch[-1]
        if model_name.startswith("ode"):
            out = model(t=batch[0], x=x, z=batch[2], event_t=batch[3], z_jump=batch[4])
        else:
            out = model(t=batch[0], x=x, z=batch[2], v=batch[3], i=batch[4], event_t=batch[5], z_jump=batch[6], v_jump=batch[7])
        pred = out[0] if isinstance(out, tuple) else out
        terms = L.masked_mse_terms(pred, x, mask.expand_as(x).contiguous() if mask.shape[-1] not in (1, x.shape[-1]) else mask)[0]
        num += terms[:x.shape[-1]].sum()
        den += mask.sum()
    return float(num / den)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="ode01", choices=sorted(HIDDEN))
    ap.add_argument("--solver", default="euler", choices=["euler", "midpoint", "rk4"], help="the scripts hard-code Euler()")
    ap.add_argument("--train-data")
    ap.add_argument("--test-data")
    ap.add_argument("--synthetic", action="store_true", help="write a small synthetic dataset first")
    ap.add_argument("--num", type=int, default=320)
    ap.add_argument("--step", type=int, default=201, help="grid points per sample (cut_length)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--save", default=None, help="directory for the TorchScript export (save_model)")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    dae = args.model.startswith("dae")
    tmp = None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        args.train_data, args.test_data = os.path.join(tmp.name, "training.npz"), os.path.join(tmp.name, "testing.npz")
        write_synthetic(args.train_data, args.num, args.step, dae, 0)
        write_synthetic(args.test_data, max(args.num // 4, 8), args.step, dae, 1)
    cls = DAE_Curves_Sample if dae else ODE_Curves_Sample
    train = datapath.ResidentDataset(cls(args.train_data, dev, num_sample=args.num, cut_length=args.step), dev)
    test = datapath.ResidentDataset(cls(args.test_data, dev), dev)
    train_loader = datapath.ResidentLoader(train, batch_size=args.batch, shuffle=True)
    test_loader = datapath.ResidentLoader(test, batch_size=max(len(test) // 10, 1), shuffle=False)
    model = build(args.model, args.solver).to(dev)
    model.solver.fused = "require"           # fail loudly rather than walk the time loop in Python
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    history = [evaluate(args.model, model, test_loader)]
    print(f"epoch 0: test x_loss_total {history[0]:.6e}")
    for epoch in range(1, args.epochs + 1):
        model.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 0
        for batch in train_loader:
            opt.zero_grad()
            loss = step_loss(args.model, model, batch)
            loss.backward()
            opt.step()
            steps += batch[0].shape[0] * (batch[0].shape[1] - 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        model.eval()
        history.append(evaluate(args.model, model, test_loader))
        print(f"epoch {epoch}: train loss {float(loss.detach()):.6e}  test x_loss_total {history[-1]:.6e}  {steps / dt / 1e6:.1f} M state-steps/s")
    if args.save:
        model.save_model(args.save)
        print("exported:", sorted(os.listdir(args.save)))
    if tmp is not None:
        tmp.cleanup()
    return history


if __name__ == "__main__":
    main()
