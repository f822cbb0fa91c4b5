"""ASDNet checkpoint -> weights.bin for asd_load_weights / host/example_track.

The reference loads a TorchScript export of ASDNet/ASDNet/ASDNet.py (`torch::jit::load`, ORBextractor.cc:457-458) or, in
training, a state_dict checkpoint (`bestmodel_c.pt`).  Neither file is part of the reference tree.  This tool reads
either kind on the CPU and writes the 7 x (conv weight [cout][cin][k][k], BN running_mean, BN running_var) f32 blobs in
layer order, which is the layout asd_load_weights takes and example_track reads:

  python tools/convert_weights.py bestmodel_c.pt weights.bin

BatchNorm is affine=False in ASDNet (ASDNet.py:336-356); a checkpoint with affine parameters is refused rather than
silently mis-folded.  BN eps is the PyTorch default 1e-5 (asd_load_weights' eps argument)."""
import sys

import numpy as np

SHAPES = [(32, 1, 3), (32, 32, 3), (64, 32, 3), (64, 64, 3), (128, 64, 3), (128, 128, 3), (128, 128, 8)]


def extract_layers(state):
    """state: mapping name -> tensor/array (a state_dict, possibly nested under 'state_dict' / 'model')"""
    for key in ("state_dict", "model", "net"):
        if key in state and hasattr(state[key], "keys"):
            state = state[key]
    items = [(k, np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v)) for k, v in state.items()]
    convs = [(k, v) for k, v in items if v.ndim == 4]
    means = [(k, v) for k, v in items if k.endswith("running_mean")]
    vars_ = [(k, v) for k, v in items if k.endswith("running_var")]
    if any(k.endswith((".weight", ".bias")) and v.ndim == 1 and not k.endswith(("running_mean", "running_var")) for k, v in items):
        raise ValueError("checkpoint has affine BatchNorm / conv bias parameters: not the ASDNet of ASDNet.py:334-356")
    if not (len(convs) == len(means) == len(vars_) == 7):
        raise ValueError(f"expected 7 conv / running_mean / running_var tensors, found {len(convs)} / {len(means)} / {len(vars_)}")
    layers = []
    for (cout, cin, k), (kc, w), (_, m), (_, v) in zip(SHAPES, convs, means, vars_):
        if w.shape != (cout, cin, k, k) or m.shape != (cout,) or v.shape != (cout,):
            raise ValueError(f"{kc}: shape {w.shape}, expected {(cout, cin, k, k)}")
        layers.append((w.astype(np.float32), m.astype(np.float32), v.astype(np.float32)))
    return layers


def write_bin(layers, path):
    with open(path, "wb") as f:
        for w, m, v in layers:
            np.ascontiguousarray(w, np.float32).tofile(f)
            np.ascontiguousarray(m, np.float32).tofile(f)
            np.ascontiguousarray(v, np.float32).tofile(f)


def read_bin(path):
    raw = np.fromfile(path, np.float32)
    layers, o = [], 0
    for cout, cin, k in SHAPES:
        nw = cout * cin * k * k
        layers.append((raw[o:o + nw].reshape(cout, cin, k, k), raw[o + nw:o + nw + cout], raw[o + nw + cout:o + nw + 2 * cout]))
        o += nw + 2 * cout
    if o != len(raw):
        raise ValueError("weights.bin has the wrong size")
    return layers


def main():
    import torch
    src, dst = sys.argv[1], sys.argv[2]
    try:
        state = torch.jit.load(src, map_location="cpu").state_dict()
    except Exception:
        state = torch.load(src, map_location="cpu")
        if hasattr(state, "state_dict"):
            state = state.state_dict()
    layers = extract_layers(state)
    write_bin(layers, dst)
    print(f"wrote {dst}: {sum(w.size + m.size + v.size for w, m, v in layers)} floats")


if __name__ == "__main__":
    main()
