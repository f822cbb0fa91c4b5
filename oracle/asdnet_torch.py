"""PyTorch-CPU restatement of the reference's descriptor path, for bench.py's cpu_baseline leg ONLY
(TEST INFRASTRUCTURE, see oracle/oracle.h).  The reference runs a TorchScript export of
ASDNet/ASDNet/ASDNet.py:334-370 through libtorch, once per pyramid level (ORBextractor.cc:1217-1231,
1125-1132); its .py cannot travel to the GPU box, so the module is restated here layer for layer and
executed the same way: eager, one forward per level, all host cores."""
import numpy as np
import torch
import torch.nn as nn


def build(layers, eps=1e-5):
    spec = [(1, 32, 3, 1, 1, True), (32, 32, 3, 1, 1, True), (32, 64, 3, 2, 1, True), (64, 64, 3, 1, 1, True),
            (64, 128, 3, 2, 1, True), (128, 128, 3, 1, 1, True), (128, 128, 8, 1, 0, False)]
    mods = []
    for (cin, cout, k, s, p, relu), (w, mean, var) in zip(spec, layers):
        conv = nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False)
        bn = nn.BatchNorm2d(cout, affine=False, eps=eps)
        conv.weight.data.copy_(torch.from_numpy(np.ascontiguousarray(w)))
        bn.running_mean.copy_(torch.from_numpy(np.ascontiguousarray(mean)))
        bn.running_var.copy_(torch.from_numpy(np.ascontiguousarray(var)))
        mods += [conv, bn] + ([nn.ReLU()] if relu else [])
        if cout == 128 and k == 3 and cin == 128:
            mods.append(nn.Dropout(0.3))
    return nn.Sequential(*mods).eval()


@torch.no_grad()
def forward(net, patches_u8):
    x = torch.from_numpy(patches_u8.astype(np.float32) * np.float32(1.0 / 255)).unsqueeze(1)
    flat = x.view(x.size(0), -1)
    mp = torch.mean(flat, dim=1)
    sp = torch.std(flat, dim=1) + 1e-7
    x = (x - mp.view(-1, 1, 1, 1)) / sp.view(-1, 1, 1, 1)
    f = net(x).view(x.size(0), -1)
    return (f / torch.sqrt(torch.sum(f * f, dim=1) + 1e-10).unsqueeze(-1)).numpy()


def describe_per_level(net, patches_u8, octaves):
    """One forward per pyramid level, like the reference."""
    out = np.empty((len(patches_u8), 128), np.float32)
    for lvl in np.unique(octaves):
        m = octaves == lvl
        out[m] = forward(net, patches_u8[m])
    return out
