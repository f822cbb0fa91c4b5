"""Is PoseOptimization deterministic?  The same 2000-edge problem solved many times (optionally beside the read-ahead extractor, which
puts ASDNet workgroups on the solver's CU); every output must be the same bits every time."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
beside = len(sys.argv) > 2 and sys.argv[2] == "beside"
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(pkg.synth.asdnet_weights(0))
img = pkg.synth.scene_frame(0)
d_img = hip.device_alloc(img.size); hip.h2d(d_img, img)
for n, seed in ((2000, 21), (1140, 5)):
    pp = pkg.synth.pose_problem(n, seed=seed, outlier_frac=0.15)
    seen = {}
    pending = 0
    for r in range(reps):
        if beside:
            while pending < 2:
                hip.extract_submit(d_img, 1241, 376, 1241); pending += 1
        got = hip.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        if beside and r % 3 == 0:
            hip.extract_wait(); pending -= 1
        key = hashlib.md5(np.asarray(got[0]).tobytes() + np.asarray(got[1]).tobytes()).hexdigest() + f" ninl={got[2]}"
        seen[key] = seen.get(key, 0) + 1
        if r == 0:
            ref = (np.array(got[0]), np.array(got[1]), got[2])
        elif seen[key] == 1 and len(seen) > 1:
            dp = np.abs(np.array(got[0]) - ref[0]).max()
            df = np.nonzero(np.array(got[1]) != ref[1])[0]
            print(f"   run {r}: pose differs by {dp:.3e}, {len(df)} outlier flags differ {df[:6]}, inliers {got[2]} vs {ref[2]}", flush=True)
    while pending:
        hip.extract_wait(); pending -= 1
    print(f"n={n} beside={beside}: {len(seen)} distinct results over {reps} runs: {sorted(seen.values(), reverse=True)[:5]}", flush=True)
