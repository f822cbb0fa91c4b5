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
from tests.test_mapping import K_KITTI, _kf_with_neighbours, _two_keyframes  # noqa: E402


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
    # ---- the per-keyframe stage in front of LocalBA, batched (round 4): LocalMapping::DoMapping's CreateNewMapPoints against nn = 20
    # neighbours and SearchInNeighbors' Fuse calls (20 neighbours + the current keyframe's points fused into each, and their points into
    # the current keyframe), each as ONE submission, against the same work as a sequence of per-pair calls
    NB = 20
    kcur, dcur, Tcur, nodes_c, has_c, nbs = _kf_with_neighbours(2000, NB, 77)
    hip.frame_set(8, kcur, dcur, BOUNDS); hip.frame_set_bow(8, nodes_c)
    for b, d in enumerate(nbs):
        hip.frame_set(9 + b, d["kps"], d["desc"], BOUNDS); hip.frame_set_bow(9 + b, d["nodes"])
    nb_args = [dict(slot=9 + b, has_mp=d["has"], F12=d["F12"], ex=d["ex"], ey=d["ey"], Tcw=d["T"], K=K_KITTI) for b, d in enumerate(nbs)]

    def per_pair():
        for b, d in enumerate(nbs):
            em, _ = hip.match_triangulate(8, 9 + b, 2000, nodes_c, d["nodes"], has_c, d["has"], d["F12"], d["ex"], d["ey"], False)
            i1 = np.nonzero(em >= 0)[0].astype(np.int32)
            hip.triangulate_pairs(8, 9 + b, i1, em[i1], Tcur, d["T"], K_KITTI, K_KITTI)
    mm, nm, _, okb = hip.create_map_points_batch(8, 2000, has_c, Tcur, K_KITTI, nb_args)
    out["create_new_map_points_batch"] = dict(gpu_ms=timeit(lambda: hip.create_map_points_batch(8, 2000, has_c, Tcur, K_KITTI, nb_args), a.reps),
                                              per_pair_calls_ms=timeit(per_pair, max(2, a.reps // 4)),
                                              note=f"{NB} neighbours x 2000 keypoints: SearchForTriangulation + triangulation, {int(nm.sum())} matches, {int(okb.sum())} new points")
    # Fuse: the current keyframe's ~1200 map points into each of the 20 neighbours, then ~2400 candidate points of the neighbours into it
    rngf = np.random.default_rng(5)
    calls, tabs, first = [], [], 0
    for c in range(NB + 1):
        slot = 9 + c if c < NB else 8
        kk = nbs[c]["kps"] if c < NB else kcur
        dd = nbs[c]["desc"] if c < NB else dcur
        Tk = nbs[c]["T"] if c < NB else Tcur
        n_mp_c = 1200 if c < NB else 2400
        srcc = rngf.integers(0, 2000, n_mp_c)
        uvc = np.stack([kk["x"][srcc], kk["y"][srcc]], 1) + rngf.uniform(-1.5, 1.5, (n_mp_c, 2)).astype(np.float32)
        Xc = backproject(Tk, K, uvc, rngf.uniform(3, 60, n_mp_c))
        Oc = -(Tk[:3, :3].astype(np.float64).T @ Tk[:3, 3].astype(np.float64))
        nc_ = Xc.astype(np.float64) - Oc
        dc_ = np.linalg.norm(nc_, axis=1)
        nc_ = (nc_ / dc_[:, None]).astype(np.float32)
        mx = (dc_ * SCALES[kk["octave"][srcc]]).astype(np.float32)
        tabs.append((np.ones(n_mp_c, np.uint8), Xc, nc_, (mx / np.float32(SCALES[7])).astype(np.float32), mx, perturbed_descriptors(dd[srcc], 0.04, 800 + c)))
        calls.append(dict(slot_kf=slot, first=first, n=n_mp_c, Tcw=Tk, K=K))
        first += n_mp_c
    cat = [np.concatenate([t[k] for t in tabs]) for k in range(6)]

    def fuse_per_call():
        for cdesc, t in zip(calls, tabs):
            hip.fuse_search(cdesc["slot_kf"], *t, cdesc["Tcw"], K, 3.0)
    bi, _ = hip.fuse_search_batch(calls, *cat, th=3.0)
    out["search_in_neighbors_fuse_batch"] = dict(gpu_ms=timeit(lambda: hip.fuse_search_batch(calls, *cat, th=3.0), a.reps),
                                                 per_call_ms=timeit(fuse_per_call, max(2, a.reps // 4)),
                                                 note=f"{NB + 1} Fuse calls, {first} candidate map points, {int((bi >= 0).sum())} found")
    hip.bank_put(0, cat[5])   # the map points' descriptors live in the bank (written when a point's descriptor changes, once per keyframe)
    rows = np.arange(first, dtype=np.int32)
    assert np.array_equal(hip.fuse_search_batch(calls, *cat[:5], rows, th=3.0)[0], bi)
    out["search_in_neighbors_fuse_batch"]["gpu_bank_rows_ms"] = timeit(lambda: hip.fuse_search_batch(calls, *cat[:5], rows, th=3.0), a.reps)
    out["per_keyframe_stage_batched_ms"] = (out["create_new_map_points_batch"]["gpu_ms"] + out["search_in_neighbors_fuse_batch"]["gpu_bank_rows_ms"] +
                                            out["distinctive_descriptor_batch"]["gpu_ms"])

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
    for k, v in out.items():
        if isinstance(v, dict):
            for kk in list(v):
                if kk.endswith("_ms"):
                    v[kk] = round(v[kk], 4)
        else:
            out[k] = round(v, 4)
    print(json.dumps({"kf_ops": out, "cpu": "oracle restatement, 1 thread", "gpu": "MI355X, wall time per C-ABI call incl. H2D/D2H"}))


if __name__ == "__main__":
    main()
