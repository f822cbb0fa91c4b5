"""Per-keyframe operations of the local-mapping thread: wall time per C-ABI call on
the GPU next to the single-thread CPU restatement (oracle) on the same inputs.  Prints one JSON line.

  python tools/kf_times.py [--voc-levels 6] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
from tests.test_matcher import BOUNDS, SCALES, _bow_nodes, _two_views, backproject, make_frame, perturbed_descriptors, pose_T  # noqa: E402
from tests.test_mapping import K_KITTI, _two_keyframes  # noqa: E402


def timeit(f, reps):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    return 1e3 * (time.perf_counter() - t) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voc-levels", type=int, default=6)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-reps", type=int, default=2)
    a = ap.parse_args()
    pkg = graft.load_package()
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    hip = pkg.capi.AsdHip(max_patches=4096)
    out = {}

    # ---- ComputeBoW: 2000 descriptors through a k=10 vocabulary
    voc = pkg.synth.vocabulary(10, a.voc_levels, seed=1)
    hip.voc_load(voc)
    V = orc.vocabulary(voc)
    kps, desc = make_frame(2000, 5)
    hip.frame_set(0, kps, desc, BOUNDS)
    g = hip.compute_bow(slot=0, n=2000, levelsup=4)
    e = V.transform(desc, levelsup=4)
    assert all(np.array_equal(x, y) for x, y in zip(g[0] + g[1], e[0] + e[1]))
    out["compute_bow"] = dict(gpu_ms=timeit(lambda: hip.compute_bow(slot=0, n=2000, levelsup=4), a.reps),
                              cpu_ms=timeit(lambda: V.transform(desc, levelsup=4), a.cpu_reps),
                              note=f"2000 descriptors, k=10 L={a.voc_levels} ({voc['n_nodes']} nodes, {voc['desc'].nbytes >> 20} MiB resident)")
    del voc

    # ---- SearchByBoW / SearchForTriangulation: one keyframe pair
    k1, d1, k2, d2, perm, F12 = _two_views(2000, 201)
    n1, n2 = _bow_nodes(d1), _bow_nodes(d2)
    hip.frame_set(1, k1, d1, BOUNDS)
    hip.frame_set(2, k2, d2, BOUNDS)
    f1, f2 = orc.frame(k1, d1, BOUNDS), orc.frame(k2, d2, BOUNDS)
    has = np.ones(2000, np.uint8)
    none = np.zeros(2000, np.uint8)
    assert np.array_equal(hip.match_bow(1, 2, 2000, n1, n2, has, 0.7, True)[0], orc.match_bow(f1, f2, n1, n2, has, 0.7, True)[0])
    out["search_by_bow"] = dict(gpu_ms=timeit(lambda: hip.match_bow(1, 2, 2000, n1, n2, has, 0.7, True), a.reps),
                                cpu_ms=timeit(lambda: orc.match_bow(f1, f2, n1, n2, has, 0.7, True), a.cpu_reps), note="2000 x 2000 keypoints, 64 nodes")
    ex, ey = 5000.0, 188.0
    gm = hip.match_triangulate(1, 2, 2000, n1, n2, none, none, F12, ex, ey, False)[0]
    assert np.array_equal(gm, orc.match_triangulate(f1, f2, n1, n2, none, none, F12, ex, ey, False)[0])
    out["search_for_triangulation"] = dict(gpu_ms=timeit(lambda: hip.match_triangulate(1, 2, 2000, n1, n2, none, none, F12, ex, ey, False), a.reps),
                                           cpu_ms=timeit(lambda: orc.match_triangulate(f1, f2, n1, n2, none, none, F12, ex, ey, False), a.cpu_reps),
                                           note="2000 x 2000 keypoints, 64 nodes, no map points yet")

    # ---- CreateNewMapPoints triangulation
    ka, kb, i1, i2, T1, T2, X, good = _two_keyframes(2000, 31)
    z = np.zeros((2000, 128), np.float32)
    hip.frame_set(3, ka, z, BOUNDS)
    hip.frame_set(4, kb, z, BOUNDS)
    gx, gok, _ = hip.triangulate_pairs(3, 4, i1, i2, T1, T2, K_KITTI, K_KITTI)
    ex_, eok, _ = orc.triangulate_pairs(ka, kb, i1, i2, T1, T2, K_KITTI, K_KITTI)
    assert np.array_equal(gx, ex_) and np.array_equal(gok, eok)
    out["triangulate"] = dict(gpu_ms=timeit(lambda: hip.triangulate_pairs(3, 4, i1, i2, T1, T2, K_KITTI, K_KITTI), a.reps),
                              cpu_ms=timeit(lambda: orc.triangulate_pairs(ka, kb, i1, i2, T1, T2, K_KITTI, K_KITTI), a.cpu_reps),
                              note=f"{len(i1)} matched pairs")

    # ---- Fuse search
    n_mp = 4000
    kc, dc = make_frame(2000, 401)
    K = np.array(pkg.synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(402)
    src = rng.integers(0, 2000, n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
    Xw = backproject(T, K, uv, rng.uniform(3, 60, n_mp))
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    normal = Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(normal, axis=1)
    normal = (normal / dist[:, None]).astype(np.float32)
    maxd = (dist * SCALES[kc["octave"][src]]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    dmp = perturbed_descriptors(dc[src], 0.04, 403)
    valid = np.ones(n_mp, np.uint8)
    hip.frame_set(5, kc, dc, BOUNDS)
    fc = orc.frame(kc, dc, BOUNDS)
    assert np.array_equal(hip.fuse_search(5, valid, Xw, normal, mind, maxd, dmp, T, K, 3.0)[0],
                          orc.fuse_search(fc, valid, Xw, normal, mind, maxd, dmp, T, K, 3.0)[0])
    out["fuse_search"] = dict(gpu_ms=timeit(lambda: hip.fuse_search(5, valid, Xw, normal, mind, maxd, dmp, T, K, 3.0), a.reps),
                              cpu_ms=timeit(lambda: orc.fuse_search(fc, valid, Xw, normal, mind, maxd, dmp, T, K, 3.0), a.cpu_reps),
                              note=f"{n_mp} map points into a 2000-keypoint keyframe")

    # ---- ComputeDistinctiveDescriptors
    obs = perturbed_descriptors(np.repeat(dc[:1], 30, 0), 0.05, 7)
    assert hip.distinctive_descriptor(obs) == orc.distinctive_descriptor(obs)
    out["distinctive_descriptor"] = dict(gpu_ms=timeit(lambda: hip.distinctive_descriptor(obs), a.reps),
                                         cpu_ms=timeit(lambda: orc.distinctive_descriptor(obs), a.cpu_reps), note="30 observations")
    # the same for a keyframe's worth of updated map points in one call
    rng = np.random.default_rng(9)
    sizes = rng.integers(2, 16, 600)
    sets = [perturbed_descriptors(np.repeat(dc[k:k + 1], n, 0), 0.05, 100 + k) for k, n in enumerate(sizes)]
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    alld = np.concatenate(sets)
    exp = [orc.distinctive_descriptor(d) for d in sets]
    assert np.array_equal(hip.distinctive_descriptor_batch(start, alld), exp)
    out["distinctive_descriptor_batch"] = dict(gpu_ms=timeit(lambda: hip.distinctive_descriptor_batch(start, alld), a.reps),
                                               cpu_ms=timeit(lambda: [orc.distinctive_descriptor(d) for d in sets], a.cpu_reps),
                                               note=f"600 map points, {int(sizes.sum())} observations, one call")
    # ---- ComputeStereoMatches: two extractor contexts, constant-disparity synthetic pair
    wide = pkg.synth.scene_frame(1, w=1241 + 96, h=376)
    left, right = np.ascontiguousarray(wide[:, 32:32 + 1241]), np.ascontiguousarray(wide[:, 44:44 + 1241])
    layers = pkg.synth.asdnet_weights(0)
    SL = pkg.capi.AsdHip(n_features=2000, max_width=1241, max_height=376)
    SR = pkg.capi.AsdHip(n_features=2000, max_width=1241, max_height=376)
    SL.load_weights(layers); SR.load_weights(layers)
    kl, dl = (x.copy() for x in SL.extract(left))
    kr, dr = (x.copy() for x in SR.extract(right))
    SL.frame_set(0, kl, dl, BOUNDS); SL.frame_set(1, kr, dr, BOUNDS)
    exl, exr = orc.extractor(2000), orc.extractor(2000)
    exl.extract(left, want_patches=False); exr.extract(right, want_patches=False)
    mb, mbf = 0.54, 0.54 * 718.856
    gu, gz, gn = SL.stereo_match(SR, 0, 1, len(kl), mb, mbf)
    eu, ez, en = exl.stereo_match(exr, kl, dl, kr, dr, mb, mbf)
    assert np.array_equal(gu, eu) and np.array_equal(gz, ez) and gn == en
    out["stereo_match"] = dict(gpu_ms=timeit(lambda: SL.stereo_match(SR, 0, 1, len(kl), mb, mbf), a.reps),
                               cpu_ms=timeit(lambda: exl.stereo_match(exr, kl, dl, kr, dr, mb, mbf), a.cpu_reps),
                               note=f"{len(kl)} x {len(kr)} keypoints, {gn} stereo matches")
    SL.close(); SR.close()
    for v in out.values():
        v["gpu_ms"] = round(v["gpu_ms"], 4)
        v["cpu_ms"] = round(v["cpu_ms"], 4)
    print(json.dumps({"kf_ops": out, "cpu": "oracle restatement, 1 thread", "gpu": "MI355X, wall time per C-ABI call incl. H2D/D2H"}))


if __name__ == "__main__":
    main()
