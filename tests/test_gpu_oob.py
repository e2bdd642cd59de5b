"""Out-of-bounds guard for the round-5 / round-6 kernels: profiles/scripts/oob_probe_*.py place every device input in turn so that it ENDS exactly at the end
of its own 32 MB allocation and run the kernels on ragged batches and grids that end inside a block; a read or write past an input is a GPU
memory access fault (the process aborts).  Run as subprocesses: a fault must fail this test, not take the test session down.
(Round 5 found one this way: a partner wave of the two-role K3f whose four trajectories all lay beyond a ragged batch read rows of a
trajectory index >= B -- invisible as long as the bytes behind the input happened to be mapped.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script", ["oob_probe_k3f.py", "oob_probe_round5.py", "oob_probe_models.py", "oob_probe_more.py", "oob_probe_round6.py", "oob_probe_generic.py"])
def test_no_kernel_touches_memory_beyond_its_inputs(script):
    env = dict(os.environ, HIP_LAUNCH_BLOCKING="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", script)], capture_output=True, text=True, timeout=900, env=env,
                         cwd=ROOT)
    tail = (res.stdout + res.stderr)[-2000:]
    assert res.returncode == 0 and "probe done" in res.stdout, tail
