"""PoseOptimization alone: kernel time per asd_pose_optimize call on a 2000-edge problem (ASD_POSE_DEBUG=1 prints passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(max_patches=4096)
for seed, n, out in ((0, 2000, 0.1), (1, 2000, 0.3), (2, 1000, 0.1)):
    pr = pkg.synth.pose_problem(n=n, seed=seed, outlier_frac=out)
    hip.pose_optimize(pr["pose"], pr["Xw"], pr["obs"], pr["info"], pr["K"])
    t = time.perf_counter()
    for _ in range(20):
        hip.pose_optimize(pr["pose"], pr["Xw"], pr["obs"], pr["info"], pr["K"])
    print(f"n={n} seed={seed}: {1e3*(time.perf_counter()-t)/20:.3f} ms wall, kernel {hip.last_stage_ms('ba'):.3f} ms", file=sys.stderr)
