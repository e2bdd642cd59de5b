"""The N>1 code path of bench.py (RCCL process group, event-table broadcast, time-chunked integrate overlapped with the all-gather)
on ONE GPU: `--force-dist` runs it with a world of size 1, THROUGH bench.py's own launcher (no WORLD_SIZE in the environment ->
bench.py starts its rank under torch.distributed.run, which is what `python3 bench.py --gpus N` does for N > 1).  Covers both
pipelined gathers (ODE, DAE) and the JSON contract fields the driver's multi-GPU run reads (integrate_only_ms, gather_only_ms,
collective, world_size_seen)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["ode01", "dae01"])
def test_bench_force_dist_runs_the_rccl_path(workload):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--grid", "301", "--batch", "512"]
    if workload == "dae01":
        cmd += ["--gather-layout", "batch"]          # RCCL with the list all_gather into strided views of the final tensors
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "launching 1 rank(s) under torch.distributed.run" in res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE JSON line, from rank 0
    d = json.loads(lines[-1])
    cfg = d["config"]
    assert d["n_gpus"] == 1 and cfg["outputs_finite"] is True
    assert cfg["world_size_seen"] == 1 and cfg["device_count"] >= 1 and cfg["rccl_version"]
    # --gather-algo auto (default, round 6): both algorithms are probed in the warm-up and the faster one runs the timed region; the
    # in-place batch layout is RCCL's own list all_gather
    assert ("rccl all_gather" in cfg["collective"]) or ("direct all-gather" in cfg["collective"])
    if workload == "dae01":
        assert cfg["gather_algo"] == "rccl"
    else:
        ch = d["multi_gpu"]["algo_choice"]
        assert ch["picked"] == cfg["gather_algo"] and set(ch["probe_gather_only_ms"]) == {"rccl", "direct"}
    assert cfg["integrate_only_ms"] > 0 and cfg["gather_only_ms"] is not None and cfg["gather_only_ms"] > 0
    assert d["roofline"]["frac"] > 0
    # round 5: the block that makes an N > 1 line self-explaining -- shard bytes, the xGMI prediction, both legs alone, what the pipeline
    # hid, the chunk count --chunks auto picked from the measured legs
    mg = d["multi_gpu"]
    assert mg["shard_bytes"] == 301 * 512 * (8 if workload == "ode01" else 10) * 4 and mg["received_bytes_per_rank"] == 0
    assert set(mg["predicted_gather_ms"]) >= {"direct_one_link_per_peer", "ring_one_link"}
    assert mg["integrate_only_ms"] > 0 and mg["gather_only_ms"] > 0 and mg["step_ms"] > 0 and "hidden_ms" in mg
    cm = mg["chunk_model"]
    assert cm is not None and 1 <= cm["chunks"] <= 16 and cm["chunks"] == mg["chunks"] and cm["integrate_only_ms"] > 0 and cm["gather_only_ms"] > 0


@pytest.mark.parametrize("extra,expect", [(["--gather-algo", "both"], "by_algo"), (["--gather-algo", "direct"], "direct"),
                                          (["--collective", "loss-only"], "loss")])
def test_bench_force_dist_gather_algorithms_and_loss_only(extra, expect):
    """Round 6: the all-gather as N - 1 point-to-point pairs (sharded.all_gather_direct) next to RCCL's own, both
    timed by --gather-algo both; and SURVEY 8(e)'s cheaper alternative, the sharded loss with scalar all-reduces only.  World size 1 here
    (the pairs degenerate to the local copy): the code path, the JSON keys and the RCCL group are what is under test."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--grid", "301",
           "--batch", "512"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    cfg, mg = d["config"], d["multi_gpu"]
    assert cfg["outputs_finite"] is True and cfg["rccl_version"]
    if expect == "by_algo":
        assert cfg["gather_algo"] == "rccl" and set(mg["by_algo"]["gather_only_ms"]) == {"rccl", "direct"} and set(mg["by_algo"]["step_ms"]) == {"rccl", "direct"}
        assert all(v > 0 for v in mg["by_algo"]["gather_only_ms"].values()) and all(v > 0 for v in mg["by_algo"]["step_ms"].values())
    elif expect == "direct":
        assert cfg["gather_algo"] == "direct" and "direct all-gather" in cfg["collective"] and mg["gather_algo"] == "direct" and mg["gather_only_ms"] > 0
    else:
        assert "all_reduce x3" in cfg["collective"] and cfg["gather_algo"] is None and cfg["gather_only_ms"] is None


def test_bench_under_an_external_launcher():
    """The contract's other spelling: python -m torch.distributed.run ... bench.py --gpus N (no self-launch)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--grid", "201", "--batch", "256"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "launching" not in res.stderr
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["outputs_finite"] is True


def test_default_bench_line_carries_the_extra_workloads():
    """BASELINE configs 3 / 4 and the scripts' shipped Euler solver ride on the default run's JSON line (reduced grid here)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    names = [e["workload"].split(":")[0] for e in d["extra"]]
    train = ["ode01 rk4 TRAIN", "dae01 rk4 TRAIN", "ode01 euler TRAIN", "dae01 euler TRAIN", "ode01 rk4 TRAIN", "dae01 rk4 TRAIN"]
    model_train = ["dae02 rk4 MODEL TRAIN", "dae02 euler MODEL TRAIN", "ode02 rk4 MODEL TRAIN"]      # whole direct_encode training steps (round 4)
    late = ["dae02 rk4", "dae02 rk4", "ode01 rk4"]       # DAE_02 forward on both routes, hidden 256 (streamed weights)
    # round 6: the direct_encode models at the scripts' argparse default --hidden 128 (forward + model training step, Euler)
    h128 = ["ode02 euler", "ode02 euler MODEL TRAIN", "dae02 euler", "dae02 euler MODEL TRAIN"]
    k0 = ["ode01_x20 rk4", "dae01_zvi16 rk4"]            # round 6: shapes without a specialisation, on the generic integrator K0 (MFMA)
    k5 = ["ode01_x20 euler TRAIN", "dae01_zvi16 euler TRAIN"]     # ... and training steps on them: K0 + K6 + the generic backward K5
    assert [n.split(" (")[0] for n in names] == ["dae01 rk4", "ode02 rk4", "ode01 euler", "dae01 euler"] + train + model_train + late + h128 + k0 + k5
    for _ in k5:
        e = d["extra"].pop()
        assert "error" not in e, e
        assert e["backward_kernel"] == "k5" and e["grads_finite"] and e["outputs_finite"] and e["roofline"]["kernel_ms_by_family"]["backward"] > 0
    for e in d["extra"][-2:]:
        assert "error" not in e, e
        assert e["kernel"] == "generic" and e["outputs_finite"] and e["gpu_vs_oracle"]["per_trajectory_rel_err"] <= 1e-5
        assert 0.15 < e["roofline"]["frac_dense"] <= e["roofline"]["frac_executed"] < 1.0      # padded tiles: executed >= dense
    d["extra"] = d["extra"][:-2]
    for e in d["extra"][-4:]:
        assert "error" not in e, e
        assert "H128" in e["workload"] and e["outputs_finite"]
    d["extra"] = d["extra"][:-4]                      # (the index-based checks below address the lines in front of them)
    for e in d["extra"][10:13]:
        assert e["grads_finite"] and e["roofline"]["flop_convention"].startswith("3 x forward")
    for e in d["extra"]:
        # round 6: ONE flop convention on every line -- frac = frac_dense (SURVEY 8(d)'s dense count; exceeds 1 where folding removes most of the
        # dense graph's work: the DAE_02 forward), frac_executed = the flops the kernel issues = pipe utilisation
        r = e["roofline"]
        assert e["outputs_finite"] and r["frac"] > 0.05 and r["kernel_ms_median"] > 0
        if "frac_dense" in r:
            assert r["frac"] == r["frac_dense"] and (r["frac_executed"] is None or 0.05 < r["frac_executed"] <= r["frac_dense"] + 1e-12)
            assert (r["frac_executed"] if r["frac_executed"] is not None else r["frac"]) < 1.0
        else:
            assert r["frac"] < 1.0
    fwd = [e for e in d["extra"] if "gpu_vs_oracle" in e]
    assert len(fwd) == 7, [e["workload"] for e in fwd]      # (+ the two hidden-128 forwards, cut off above)      # every forward line is checked against the oracle
    for e in fwd:
        assert "error" not in e["gpu_vs_oracle"], e["gpu_vs_oracle"]
        assert e["gpu_vs_oracle"]["per_trajectory_rel_err"] <= 1e-5, (e["workload"], e["gpu_vs_oracle"])
    assert d["roofline"]["frac"] == d["roofline"]["frac_dense"] and 0.3 < d["roofline"]["frac_executed"] < d["roofline"]["frac_dense"]
    assert "H256" in d["extra"][-1]["workload"] and d["extra"][-1]["kernel"] == "mfma" and d["extra"][-1]["roofline"]["frac"] > 0.5
    assert "K3g" in d["extra"][-2]["workload"] and "default route" in d["extra"][-3]["workload"]
    # round 4: the training steps (row f1) ride on the driver's clock too -- hidden 64 (RK4, Euler) and the scripts' --hidden 128
    tr = d["extra"][4:10]
    assert [("H64" in e["workload"], "H128" in e["workload"]) for e in tr] == [(True, False)] * 4 + [(False, True)] * 2
    for e in tr:
        fam = e["roofline"]["kernel_ms_by_family"]
        assert e["grads_finite"] and fam["forward"] > 0 and fam["backward"] > fam["forward"] and e["host_enqueue_ms"] > 0
        assert e["roofline"]["flop_convention"].startswith("3 x forward")
    assert all(e["saved_bytes"] > 0 for e in tr)       # every hidden-64 / 128 training step reads saved activations (DAE_01 Euler too since round 4)
    assert [e["roofline"]["bound"] for e in d["extra"][:4]] == ["mfma", "valu_fp32", "mfma", "mfma"]      # K3f issues no MFMA
    assert d["roofline"]["kernel_ms_median"] > 0
