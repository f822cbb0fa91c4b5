"""tools/convert_weights.py: ASDNet checkpoints (state_dict or TorchScript) -> the weights.bin layout of asd_load_weights."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("convert_weights", os.path.join(ROOT, "tools", "convert_weights.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_state_dict_and_torchscript_checkpoints(synth, tmp_path):
    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    from oracle import asdnet_torch
    layers = synth.asdnet_weights(3)
    net = asdnet_torch.build(layers)
    torch.save(net.state_dict(), tmp_path / "sd.pt")
    torch.save({"epoch": 7, "state_dict": net.state_dict()}, tmp_path / "ckpt.pt")
    torch.jit.trace(net, torch.zeros(1, 1, 32, 32)).save(str(tmp_path / "script.pt"))
    tool = _tool()
    for name in ("sd.pt", "ckpt.pt", "script.pt"):
        out = tmp_path / (name + ".bin")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), str(tmp_path / name), str(out)])
        got = tool.read_bin(out)
        for (w, m, v), (gw, gm, gv) in zip(layers, got):
            np.testing.assert_array_equal(gw, w)
            np.testing.assert_array_equal(gm, m)
            np.testing.assert_array_equal(gv, v)
    # the size example_track checks: 7 x (cout*cin*k*k + 2*cout) floats
    assert os.path.getsize(tmp_path / "sd.pt.bin") == 4 * sum(co * ci * k * k + 2 * co for co, ci, k in tool.SHAPES)


def test_refuses_affine_batchnorm(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.nn as nn
    tool = _tool()
    net = nn.Sequential(nn.Conv2d(1, 32, 3, bias=False), nn.BatchNorm2d(32, affine=True))
    with pytest.raises(ValueError):
        tool.extract_layers(net.state_dict())
