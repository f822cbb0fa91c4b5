"""LocalBundleAdjustment solved repeatedly beside a busy read-ahead extractor: every output must be the same bits every time
(the companion of pose_determinism.py for the LocalBA kernels, which share their CUs with ASDNet workgroups in the bench)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(pkg.synth.asdnet_weights(0))
img = pkg.synth.scene_frame(0)
d_img = hip.device_alloc(img.size); hip.h2d(d_img, img)
prob = pkg.synth.ba_problem(seed=1)
seen, pending = {}, 0
for r in range(reps):
    while pending < 2:
        hip.extract_submit(d_img, 1241, 376, 1241); pending += 1
    p = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in prob.items()}
    got = hip.local_ba(p)
    if r % 2 == 0:
        hip.extract_wait(); pending -= 1
    key = hashlib.md5(b"".join(np.ascontiguousarray(got[k]).tobytes() for k in ("poses", "points", "edge_chi2", "edge_outlier1"))).hexdigest()
    seen[key] = seen.get(key, 0) + 1
while pending:
    hip.extract_wait(); pending -= 1
print(f"LocalBA beside the extractor: {len(seen)} distinct results over {reps} runs: {sorted(seen.values(), reverse=True)[:5]}")
