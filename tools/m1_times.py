"""asd_match_project_frame / _points on an idle device and beside the read-ahead extractor: device time and the k_resolve
phase stamps (ASD_TIMING=1 prints them every 200 calls)."""
import os
import sys

os.environ.setdefault("ASD_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
synth = pkg.synth
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(synth.asdnet_weights(0))
B = (0.0, 1241.0, 0.0, 376.0)
k0, d0 = hip.extract(synth.scene_frame(10)); k0, d0 = k0.copy(), d0.copy()
k1, d1 = hip.extract(synth.scene_frame(11)); k1, d1 = k1.copy(), d1.copy()
hip.frame_set(0, k1, d1, B)
hip.frame_set(1, k0, d0, B)
K = np.array(synth.KITTI_K, np.float32)
T = np.eye(4, dtype=np.float32)
z = 1.003
uv = np.stack([(k0["x"] - 620.5) * z + 620.5 - 3 * z, (k0["y"] - 188.0) * z + 188.0 - 0.2 * z], 1).astype(np.float32)
Xw = np.stack([(uv[:, 0] - K[2]) / K[0] * 20, (uv[:, 1] - K[3]) / K[1] * 20, np.full(len(uv), 20.0)], 1).astype(np.float32)
has = np.ones(len(k0), np.uint8)
busy = len(sys.argv) > 1 and sys.argv[1] == "busy"
im = synth.scene_frame(5)
p = hip.device_alloc(im.nbytes)
hip.h2d(p, im)
for it in range(400):
    if busy:
        hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
    m, n = hip.match_project_frame(0, 1, len(k1), has, Xw, d0, T, K, 15.0, True)
    if busy:
        hip.extract_wait()
print("matches", n, "of", len(k0))
hip.close()
