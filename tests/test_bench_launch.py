"""`python bench.py --gpus N` must start its own ranks (the driver calls it without a launcher): on this GPU-less container the two
spawned ranks get as far as the device check and fail THERE -- not on argv, not on the rendezvous."""
import os
import subprocess
import sys

import torch
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-container check; the GPU box runs tests/test_gpu_bench_dist.py")
def test_bench_gpus_2_launches_two_ranks_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert "launching 2 rank(s) under torch.distributed.run" in res.stderr
    # the ranks ran main() as far as the device check (usually both print it; torchrun's agent may terminate the second one the moment the
    # first has failed, before it gets there: one message is enough to show the launch itself worked)
    assert 1 <= res.stderr.count("bench.py needs a HIP device") <= 2, res.stderr[-3000:]
    assert "needs `python -m torch.distributed.run" not in res.stderr


def test_launched_ranks_keep_the_world_the_launcher_gave_them():
    """Under a launcher (WORLD_SIZE set) bench.py never re-launches; a --gpus that disagrees with the launcher's world size is an error
    (a line that says n_gpus = 8 must have run on 8 ranks), an agreeing one goes on to the device check."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    if torch.cuda.is_available():
        pytest.skip("needs the GPU-less container (the rank would go on to the rendezvous)")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert res.returncode != 0 and "launching" not in res.stderr and "--gpus 8 but the launcher started WORLD_SIZE=2" in res.stderr
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert res.returncode != 0 and "launching" not in res.stderr and "bench.py needs a HIP device" in res.stderr


def test_multi_gpu_report_keys_and_model():
    """The N > 1 block of the bench line (pure host arithmetic): shard bytes, the xGMI prediction band, the two legs alone, what the
    pipeline hid, the chunk model of --chunks auto."""
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["ode01"])
    B, T, N = 4096, 1001, 8
    cm = {"integrate_only_ms": 3.3, "gather_only_ms": 1.2, "per_chunk_overhead_ms": 0.05, "chunks": 5, "predicted_total_ms": 3.3 + 1.2 / 5 + 0.25}
    r = bench.multi_gpu_report(w, N, B, T, step_ms=3.7, integrate_ms=3.3, gather_ms=1.2, chunks=5, pipelined=True, chunk_model=cm)
    assert r["shard_bytes"] == T * B * 8 * 4 and r["received_bytes_per_rank"] == 7 * r["shard_bytes"]
    band = r["predicted_gather_ms"]
    assert abs(band["direct_one_link_per_peer"] - r["shard_bytes"] / 153e9 * 1e3) < 1e-9
    assert abs(band["ring_one_link"] - 7 * band["direct_one_link_per_peer"]) < 1e-9
    assert abs(r["serial_ms"] - 4.5) < 1e-12 and abs(r["hidden_ms"] - 0.8) < 1e-12 and abs(r["hidden_frac_of_shorter_leg"] - 0.8 / 1.2) < 1e-12
    assert r["chunks"] == 5 and r["pipelined"] and r["chunk_model"] is cm
    for k in ("integrate_only_ms", "gather_only_ms", "step_ms", "gather_achieved_GBs_per_rank"):
        assert k in r
    assert r["gather_algo"] is None and r["by_algo"] is None
    both = {"gather_only_ms": {"rccl": 1.2, "direct": 0.9}, "step_ms": {"rccl": 3.7, "direct": 3.5}}
    r2 = bench.multi_gpu_report(w, N, B, T, 3.7, 3.3, 1.2, 5, True, cm, "rccl", both)
    assert r2["gather_algo"] == "rccl" and r2["by_algo"] is both
    d = bench.multi_gpu_report(dict(bench.WORKLOADS["dae01"]), 2, 4096, 1001, 5.0, 4.6, None, None, False, None)
    assert d["shard_bytes"] == 1001 * 4096 * (8 + 2) * 4 and "serial_ms" not in d
