"""world_size-2 `gloo` test of the multi-GPU path on CPU: shard -> integrate locally -> all-gather == unsharded.

The HIP kernel cannot run here, so the per-rank integrate is the CPU oracle (tests may use it); what is under test is
py_psnode_amd/sharded.py: slicing, the broadcast of the event table decided by GLOBAL trajectory 0, and the gather."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _oracle_local(method, de, t, x, z, a0, z_jump=None, input_true_x=False, event_idx=None, out=None):
    """Oracle driven by an explicit per-step event table (what the sharded wrapper hands every rank)."""
    from oracle import psnode_oracle as O
    xs = torch.zeros(t.shape[0], x.shape[1], x.shape[2])
    cur = x[0]
    xs[0] = cur
    for k in range(t.shape[0] - 1):
        zk = z[k]
        if event_idx is not None and int(event_idx[k]) >= 0:
            zk = z_jump[:, int(event_idx[k])]
        src = x[k] if input_true_x else cur
        cur, _ = O.step(method, lambda xx: O.de_rhs(de, xx, (zk,), a0), t[k], t[k + 1] - t[k], t[k + 1], src)
        xs[k + 1] = cur
    if out is not None:
        out.copy_(xs)
        return out
    return xs


def _torch_masked_mse(pred, target, mask, col_weight=None, inv_norm=None, scale=1.0, t0_coef=0.0):
    """CPU stand-in for py_psnode_amd.loss.masked_mse (same contract: (total, [per-column..., t0 term, total])), in torch ops."""
    D = pred.shape[2]
    se = (pred - target) ** 2
    w = torch.ones(D) if col_weight is None else torch.as_tensor(col_weight, dtype=torch.float32)
    cols = torch.sum(se * (mask if mask is not None else 1.0), dim=(0, 1)) * w * scale * (inv_norm if inv_norm is not None else 1.0)
    t0 = t0_coef * torch.sum(se[:, 0, :])
    tot = cols.sum() + t0
    return tot, torch.cat((cols, t0.view(1), tot.view(1)))


def _check_sharded_loss(rank, world, sharded):
    """Sharded loss + gradient all-reduce == the unsharded oracle loss and its autograd gradients (DAE-style weights + t0 term)."""
    from oracle import psnode_oracle as O
    g = torch.Generator().manual_seed(7)
    B, Tn, D = 10, 9, 8
    lin = torch.nn.Linear(D, D)
    with torch.no_grad():
        lin.weight.copy_(0.3 * torch.randn(D, D, generator=g)); lin.bias.copy_(0.1 * torch.randn(D, generator=g))
    inp, x = torch.randn(B, Tn, D, generator=g), torch.randn(B, Tn, D, generator=g)
    mask = (torch.rand(B, Tn, 1, generator=g) > 0.3).float()
    i_dummy = torch.zeros(B, Tn, 1)
    ref = O.dae_loss(lin(inp), x, i_dummy, i_dummy, mask)          # x part: weights [1,10,1,...] + the t=0 term
    ref_x = ref[1] + ref[3]
    gw, gb = torch.autograd.grad(ref_x, (lin.weight, lin.bias))
    lo, hi = sharded.shard_bounds(B, rank, world)
    lin.zero_grad()
    share, terms = sharded.masked_mse_sharded(lin(inp[lo:hi]), x[lo:hi], mask[lo:hi], col_weight=[1.0, 10.0] + [1.0] * (D - 2),
                                             t0_weight=1.0, global_batch=B, local_fn=_torch_masked_mse)
    share.backward()
    sharded.all_reduce_param_grads(lin.parameters())
    assert abs(float(terms[D + 1]) - float(ref_x)) <= 1e-5 * abs(float(ref_x)), (float(terms[D + 1]), float(ref_x))
    assert float((lin.weight.grad - gw).abs().max()) <= 1e-5 * float(gw.abs().max())
    assert float((lin.bias.grad - gb).abs().max()) <= 1e-5 * float(gb.abs().max())


def _oracle_dae_local(method, de, ae, x_init, t, x, z, v, i, a0, z_jump=None, v_jump=None, event_idx=None, out=None):
    """DAE oracle driven by an explicit per-step event table (no teacher forcing)."""
    from oracle import psnode_oracle as O
    Tn, B = t.shape[0], t.shape[1]
    xs, is_ = torch.zeros(Tn, B, x_init.shape[-1]), torch.zeros(Tn, B, i.shape[-1])
    cx = x_init
    ci = O.ae_rhs(ae, cx, z[0], v[0], a0)
    xs[0], is_[0] = cx, ci
    for k in range(Tn - 1):
        zk, vk = z[k], v[k]
        if event_idx is not None and int(event_idx[k]) >= 0:
            zk, vk = z_jump[:, int(event_idx[k])], v_jump[:, int(event_idx[k])]
            ci = O.ae_rhs(ae, cx, zk, vk, a0)
        cx, _ = O.step(method, lambda xx: O.de_rhs(de, xx, (zk, vk, ci), a0), t[k], t[k + 1] - t[k], t[k + 1], cx)
        ci = O.ae_rhs(ae, cx, z[k + 1], v[k + 1], a0)
        xs[k + 1], is_[k + 1] = cx, ci
    if out is not None:
        out[0].copy_(xs); out[1].copy_(is_)
        return out
    return xs, is_


def _check_dae_pipelined(rank, world, sharded):
    """integrate_dae_pipelined (time chunks restarted from xs[s], is recomputed by the launch) == the reference's one-shot
    integrate_DAE golden (G3, events on), xs and is reassembled from the chunk-major gathers."""
    from helpers import T, layers, load, tm
    d = load("g3_dae.npz")
    de, ae = layers(d, "de__x_dot"), layers(d, "ae__i_calculator")
    t, z, v, i = tm(d["t"]), tm(d["z"]), tm(d["v"]), tm(d["i"])
    xi, a0, ev, zj, vj = T(d["x_init"]), T(d["all_initial"]), T(d["event_t"]), T(d["z_jump"]), T(d["v_jump"])
    lo, hi = sharded.shard_bounds(t.shape[1], rank, world)
    tab = sharded.broadcast_event_table(t[:, lo:hi], ev[lo:hi], table_fn=_table)
    (xl, il), (gx, gi) = sharded.integrate_dae_pipelined("rk4", de, ae, xi[lo:hi], t[:, lo:hi], z[:, lo:hi], v[:, lo:hi], i[:, lo:hi],
                                                          a0[lo:hi], event_idx=tab, z_jump=zj[lo:hi], v_jump=vj[lo:hi], chunks=4,
                                                          local_fn=_oracle_dae_local)
    X, I = sharded.assemble(gx, t.shape[0]), sharded.assemble(gi, t.shape[0])
    rx, ri = T(d["rk4_tx0_ti0_ev1_x"]), T(d["rk4_tx0_ti0_ev1_i"])
    assert float((X - rx).abs().max()) <= 2e-6 and float((I - ri).abs().max()) <= 2e-6
    assert torch.equal(xl, X[:, lo:hi]) and torch.equal(il, I[:, lo:hi])
    # layout="batch": both outputs gathered straight into their final [T, G*Bl, D] tensors
    _, (fx, fi) = sharded.integrate_dae_pipelined("rk4", de, ae, xi[lo:hi], t[:, lo:hi], z[:, lo:hi], v[:, lo:hi], i[:, lo:hi], a0[lo:hi],
                                                  event_idx=tab, z_jump=zj[lo:hi], v_jump=vj[lo:hi], chunks=4, local_fn=_oracle_dae_local,
                                                  layout="batch", want_local=False)
    assert torch.equal(fx, X) and torch.equal(fi, I)
    # the direct gather: same rows
    _, (dx, di) = sharded.integrate_dae_pipelined("rk4", de, ae, xi[lo:hi], t[:, lo:hi], z[:, lo:hi], v[:, lo:hi], i[:, lo:hi], a0[lo:hi],
                                                  event_idx=tab, z_jump=zj[lo:hi], v_jump=vj[lo:hi], chunks=4, local_fn=_oracle_dae_local,
                                                  want_local=False, algo="direct")
    assert torch.equal(sharded.assemble(dx, t.shape[0]), X) and torch.equal(sharded.assemble(di, t.shape[0]), I)


def _table(t, event_t):
    from oracle import psnode_oracle as O
    return torch.tensor(O.event_step_table(t, event_t), dtype=torch.int32)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import T, layers, load, tm
        from py_psnode_amd import sharded
        d = load("g2_ode.npz")
        de = layers(d, "de__x_dot")
        t, x, z = tm(d["t"]), tm(d["x"]), tm(d["z"])
        a0, ev, zj = T(d["all_initial"]), T(d["event_t"]), T(d["z_jump"])
        B = t.shape[1]
        lo, hi = sharded.shard_bounds(B, rank, world)
        # make rank 1's local trajectory 0 carry a DIFFERENT clock: the table must still come from global trajectory 0
        tl = t[:, lo:hi].clone()
        if rank == 1:
            tl[:, 0] = tl[:, 0] + 0.123
        out = sharded.integrate_ode_sharded("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_t=ev[lo:hi], z_jump=zj[lo:hi],
                                            local_fn=_oracle_local, table_fn=_table)
        ref = T(d["rk4_events"]).clone()
        # time-chunked + pipelined gather: same numbers, reassembled from the chunk-major buffers
        tab = sharded.broadcast_event_table(tl, ev[lo:hi], table_fn=_table)
        xs_l, gathered = sharded.integrate_ode_pipelined("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_idx=tab,
                                                         z_jump=zj[lo:hi], chunks=3, local_fn=_oracle_local)
        assert torch.equal(sharded.assemble(gathered, tl.shape[0]), out), "pipelined gather differs from the one-shot gather"
        # layout="batch": every chunk's gather lands directly in the final [T, G*Bl, xd] tensor -- no assemble() pass
        _, full = sharded.integrate_ode_pipelined("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_idx=tab, z_jump=zj[lo:hi],
                                                  chunks=3, local_fn=_oracle_local, layout="batch", want_local=False)
        assert isinstance(full, torch.Tensor) and full.is_contiguous() and torch.equal(full, out), "in-place gather differs"
        # the DIRECT gather (round 6: N - 1 point-to-point pairs per chunk, every peer's shard on its own link) == RCCL's own all-gather ==
        # the unsharded golden, one-shot and pipelined; layout="batch" belongs to c10d's list all_gather and is refused with it
        out_d = sharded.integrate_ode_sharded("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_t=ev[lo:hi], z_jump=zj[lo:hi],
                                              local_fn=_oracle_local, table_fn=_table, algo="direct")
        assert torch.equal(out_d, out), "direct gather differs from all_gather_into_tensor"
        _, gathered_d = sharded.integrate_ode_pipelined("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_idx=tab, z_jump=zj[lo:hi],
                                                        chunks=3, local_fn=_oracle_local, algo="direct")
        assert torch.equal(sharded.assemble(gathered_d, tl.shape[0]), out), "pipelined direct gather differs"
        try:
            sharded.integrate_ode_pipelined("rk4", de, tl, x[:, lo:hi], z[:, lo:hi], a0[lo:hi], event_idx=tab, z_jump=zj[lo:hi], chunks=3,
                                            local_fn=_oracle_local, layout="batch", algo="direct")
            raise AssertionError("layout='batch' with the direct gather was accepted")
        except ValueError as e:
            assert "direct" in str(e)
        assert sharded.chunk_bounds(1001, 4) == [0, 250, 500, 751, 1001] and sharded.chunk_bounds(2, 4) == [0, 1, 2]
        assert torch.equal(xs_l, out[:, lo:hi]), "local rows of the pipelined run differ from this rank's slice of the gather"
        _check_dae_pipelined(rank, world, sharded)
        # unequal shards must raise on every rank instead of hanging inside the collective -- also when ONE rank has seen its size
        # before (full batches of 4 + 4, then a ragged last batch of 4 + 3: round-2 ADVICE, a per-rank memo of the sizes let rank 0
        # skip the check's all-reduce while rank 1 entered it)
        full = sharded.all_gather_batch(torch.full((3, 4, 2), float(rank)))
        assert full.shape == (3, 8, 2) and torch.equal(full[:, :4], torch.zeros(3, 4, 2)) and torch.equal(full[:, 4:], torch.ones(3, 4, 2))
        for ragged in (4 - rank, 4 + rank):
            try:
                sharded.all_gather_batch(torch.zeros(3, ragged, 2))
                raise AssertionError("unequal shards were accepted")
            except ValueError as e:
                assert "equal shards" in str(e)
        again = sharded.all_gather_batch(torch.full((3, 4, 2), float(rank)), check_shards=False)    # the ranks are still in lock-step
        assert torch.equal(again, full)
        if rank == 0:
            # trajectory `B/2` (rank 1's first) was integrated with the shifted clock but identical dt -> same result
            q.put((tuple(out.shape), float((out - ref).abs().max())))
        _check_sharded_loss(rank, world, sharded)
        bounds = [sharded.shard_bounds(10, r, 3) for r in range(3)]
        assert bounds == [(0, 4), (4, 7), (7, 10)]
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_then_gather_equals_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, "worker failed"
    shape, err = q.get(timeout=10)
    assert shape == (101, 32, 8)
    assert err <= 2e-6


def _worker_direct(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from py_psnode_amd import sharded
        g = torch.Generator().manual_seed(100 + rank)
        shard = torch.randn(5, 3, 2, generator=g)
        want = torch.empty(world * 5, 3, 2)
        dist.all_gather_into_tensor(want, shard)
        got = torch.full((world * 5, 3, 2), float("nan"))
        assert sharded.all_gather_direct(got, shard) is None
        assert torch.equal(got, want), "direct gather (blocking) differs from all_gather_into_tensor"
        got2 = torch.full((world * 5, 3, 2), float("nan"))
        sharded.all_gather_direct(got2, shard, async_op=True).wait()
        assert torch.equal(got2, want)
        full = sharded.all_gather_batch(shard, algo="direct")
        assert torch.equal(full, sharded.all_gather_batch(shard))
        try:
            sharded.all_gather_direct(torch.empty(world * 5 + 1, 3, 2), shard)
            raise AssertionError("a mis-sized destination was accepted")
        except ValueError:
            pass
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_direct_gather_world_size_3():
    """all_gather_direct at an odd world size (round d pairs r with r + d and r - d: every peer once, no self-exchange): bit-equal to
    all_gather_into_tensor, blocking and async."""
    ctx = mp.get_context("spawn")
    port = 30100 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker_direct, args=(r, 3, port)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "worker failed"
