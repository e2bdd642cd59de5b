"""The end-to-end example (examples/train_direct.py: resident loader -> fused model -> fused loss -> fused backward -> Adam ->
evaluation -> TorchScript export) runs for all four shipped model configurations and the test loss goes down."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", ["ode01", "ode02", "dae01", "dae02"])
def test_example_training_loop(model, tmp_path):
    import train_direct
    out = tmp_path / "export"
    hist = train_direct.main(["--model", model, "--synthetic", "--num", "96", "--step", "41", "--batch", "32", "--epochs", "4", "--lr", "3e-3",
                              "--save", str(out)])
    assert len(hist) == 5 and all(h == h for h in hist), hist
    assert hist[-1] < hist[0], f"test loss did not go down: {hist}"
    assert "de_func.pt" in os.listdir(out)
