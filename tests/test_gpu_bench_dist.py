"""The N>1 code path of bench.py (RCCL process group, event-table broadcast, time-chunked integrate overlapped with the all-gather)
on ONE GPU: `--force-dist` runs it with a world of size 1.  Covers both pipelined gathers (ODE, DAE) and the JSON contract fields
the driver's multi-GPU run reads (integrate_only_ms, gather_only_ms, collective)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["ode01", "dae01"])
def test_bench_force_dist_runs_the_rccl_path(workload):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + (os.getpid() % 300) + (1 if workload == "dae01" else 0)),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--grid", "301", "--batch", "512"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    cfg = d["config"]
    assert d["n_gpus"] == 1 and cfg["outputs_finite"] is True
    assert "time chunks overlapped" in cfg["collective"]
    assert cfg["integrate_only_ms"] > 0 and cfg["gather_only_ms"] is not None and cfg["gather_only_ms"] > 0
    assert d["roofline"]["frac"] > 0
