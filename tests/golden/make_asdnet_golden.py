#!/usr/bin/env python3
"""Generate tests/golden/asdnet_golden.npz from the REFERENCE's own ASDNet class.

Runs only in the build container (needs /root/reference).  The reference module
ASDNet/ASDNet/ASDNet.py is imported as-is; cv2 / torchvision are not installed here and
are only used by its training / data code (ASDNet.py:17,26-27,234, Utils.py:4), so they
are stubbed in sys.modules.  The trained weights are absent from the reference tree
(.MISSING_LARGE_BLOBS:1-2): seeded weights and BN running stats from
asd-slam_amd/synth.py are loaded into the reference module instead.

Fixture = inputs (u8 patches) + expected outputs (f32 descriptors); the weights are
regenerated from the seed on both sides, so the fixture stays small.
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/ASDNet/ASDNet"


def load_synth():
    spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "asd-slam_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def import_reference_asdnet():
    import torch  # noqa: F401
    sys.dont_write_bytecode = True
    for name in ("cv2", "torchvision", "torchvision.datasets", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.datasets"].PhotoTour = type("PhotoTour", (), {})
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = ["ASDNet.py", "--no-cuda"]
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)  # module-level os.makedirs(args.log_dir) (ASDNet.py:226-227)
    try:
        import ASDNet as ref
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    return ref


def main():
    import torch
    synth = load_synth()
    ref = import_reference_asdnet()
    torch.manual_seed(0)
    net = ref.ASDNet()
    layers = synth.asdnet_weights(seed=0)
    convs = [m for m in net.features if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in net.features if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(convs) == 7 and len(bns) == 7
    for (w, mean, var), c, b in zip(layers, convs, bns):
        assert tuple(c.weight.shape) == w.shape
        c.weight.data.copy_(torch.from_numpy(w))
        b.running_mean.copy_(torch.from_numpy(mean))
        b.running_var.copy_(torch.from_numpy(var))
    net.eval()
    patches = synth.random_patches(64, seed=1)
    # ORBextractor.cc:1125-1128: u8 -> f32 / 255, [n,32,32,1] -> [n,1,32,32]
    x = torch.from_numpy(patches.astype(np.float32) * np.float32(1.0 / 255)).unsqueeze(1)
    torch.set_num_threads(1)
    with torch.no_grad():
        y = net(x).numpy().astype(np.float32)
        # per-layer activations of 2 patches, to localise a mismatch
        a = net.input_norm(x[:4])
        acts = [a.numpy().copy()]
        for m in net.features:
            a = m(a)
            if isinstance(m, (torch.nn.ReLU,)):
                acts.append(a.numpy().copy())
        acts.append(a.numpy().copy())
    out = os.path.join(ROOT, "tests", "golden", "asdnet_golden.npz")
    np.savez_compressed(out, weight_seed=0, patches=patches, desc=y,
                        act_norm=acts[0], act_l1=acts[1][:, :, ::8, ::8], act_l6=acts[6], act_l7=acts[7])
    print("wrote", out, y.shape, "row norms", np.linalg.norm(y, axis=1)[:4])


if __name__ == "__main__":
    main()
