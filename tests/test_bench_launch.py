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
    assert res.stderr.count("bench.py needs a HIP device") == 2, res.stderr[-3000:]      # both ranks ran main()
    assert "needs `python -m torch.distributed.run" not in res.stderr


def test_launched_ranks_keep_the_world_the_launcher_gave_them():
    """Under a launcher (WORLD_SIZE set) bench.py never re-launches, whatever --gpus says."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    if torch.cuda.is_available():
        pytest.skip("needs the GPU-less container (the rank would go on to the rendezvous)")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert res.returncode != 0 and "launching" not in res.stderr and "bench.py needs a HIP device" in res.stderr
