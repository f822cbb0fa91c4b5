#!/usr/bin/env python3
"""Generate tests/golden/ba_golden.npz with the REFERENCE's own vendored g2o, compiled in place from
/root/reference/src/g2o_catkin by oracle/ref_g2o/Makefile (outputs in oracle/_ref/, never committed).

Runs only in the build container.  Problems come from asd-slam_amd/synth.py (seeded); the fixture stores the
generator arguments, the inputs (so a generator change cannot silently move the goalposts) and g2o's outputs
for Optimizer::PoseOptimization's and Optimizer::LocalBundleAdjustment's schedules (oracle/ref_g2o/driver.cpp).
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

BA_CASES = [
    dict(n_free=5, n_fixed=3, n_points=300, seed=1),
    dict(n_free=7, n_fixed=1, n_points=500, seed=5, outlier_frac=0.1, pose_sigma=0.05),
    dict(n_free=3, n_fixed=0, n_points=120, seed=9, obs_per_point=3, point_sigma=0.2),
]
POSE_CASES = [
    dict(n=300, seed=2),
    dict(n=250, seed=4, outlier_frac=0.25, pose_sigma=0.05),
    dict(n=40, seed=6, outlier_frac=0.0),
    dict(n=8, seed=7),
]


def main():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle", "ref_g2o")])
    synth = g.load_package().synth
    ref = g.load_oracle().RefG2O()
    out = {"ba_cases": json.dumps(BA_CASES), "pose_cases": json.dumps(POSE_CASES)}
    for i, kw in enumerate(BA_CASES):
        prob = synth.ba_problem(**kw)
        if kw.get("n_fixed", 1) == 0:
            prob["fixed"][0] = 1  # KF 0 is always fixed (Optimizer.cc:490)
        res = ref.local_ba(prob)
        for k, v in prob.items():
            out[f"ba{i}_in_{k}"] = v
        for k, v in res.items():
            out[f"ba{i}_out_{k}"] = np.asarray(v)
        print("ba", i, "E", len(prob["e_point"]), res["chi2_first"], res["chi2_second"], res["iters_first"],
              res["iters_second"], "outliers", int(res["edge_outlier1"].sum()))
    for i, kw in enumerate(POSE_CASES):
        pp = synth.pose_problem(**kw)
        pose, outlier, ninl = ref.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        for k, v in pp.items():
            out[f"pose{i}_in_{k}"] = v
        out[f"pose{i}_out_pose"] = pose
        out[f"pose{i}_out_outlier"] = outlier
        out[f"pose{i}_out_ninl"] = np.int32(ninl)
        print("pose", i, ninl, "of", kw["n"])
    path = os.path.join(ROOT, "tests", "golden", "ba_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
