"""Relocalisation / loop-closing variants of the projection searches:
SearchByProjection(Frame, KeyFrame, set, th, ORBdist), SearchByProjection(KeyFrame, Scw, ...), Fuse(KeyFrame, Scw, ...),
SearchBySim3 -- HIP vs the CPU restatement, bit-exact ids and distances."""
import numpy as np
import pytest

from tests.test_matcher import BOUNDS, SCALES, backproject, make_frame, perturbed_descriptors, pose_T


def _map_points_for(kc, dc, T, K, n_mp, seed, jitter=1.5, desc_sigma=0.04):
    """n_mp map points that project near random keypoints of the frame (kc, dc) under pose T"""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, len(kc), n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-jitter, jitter, (n_mp, 2)).astype(np.float32)
    uv[: n_mp // 10] += 3000                                        # out of the image
    Xw = backproject(T, K, uv, rng.uniform(3, 60, n_mp))
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    PO = Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = PO / dist[:, None] + rng.normal(0, 0.35, PO.shape)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    maxd = (dist * SCALES[kc["octave"][src]] * rng.uniform(0.9, 1.1, n_mp)).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    desc = perturbed_descriptors(dc[src], desc_sigma, seed + 1)
    valid = (rng.uniform(size=n_mp) < 0.9).astype(np.uint8)
    return dict(src=src, Xw=Xw, normal=normal, mind=mind, maxd=maxd, desc=desc, valid=valid)


def _sim3(T, s):
    S = T.copy()
    S[:3, :] = (np.float32(s) * T[:3, :]).astype(np.float32)
    return S


# ------------------------------------------------------------------ CPU: oracle known answers
def test_oracle_reloc_projection_recovers_sources(oracle, synth):
    kc, dc = make_frame(1500, 501)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    mp = _map_points_for(kc, dc, T, K, 1200, 502)
    occ = np.zeros(len(kc), np.uint8)
    occ[::7] = 1
    ang = (kc["angle"][mp["src"]] + np.random.default_rng(1).normal(0, 2, 1200)).astype(np.float32) % np.float32(360)
    m, n = oracle.match_project_keyframe(oracle.frame(kc, dc, BOUNDS), mp["valid"], mp["Xw"], mp["mind"], mp["maxd"], mp["desc"], ang, occ,
                                         T, K, 10.0, 1.0, True)
    ok = m >= 0
    assert n == ok.sum() and n > 300
    assert not (ok & (occ == 1)).any()                       # occupied keypoints are never taken
    assert (mp["valid"][m[ok]] == 1).all()
    assert (mp["src"][m[ok]] == np.nonzero(ok)[0]).mean() > 0.95


def test_oracle_sim3_variants_scale_invariant(oracle, synth):
    """Scw = s * Tcw must give the same projections as Tcw for any s > 0 (the searches divide the scale out)"""
    kc, dc = make_frame(1200, 511)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    mp = _map_points_for(kc, dc, T, K, 900, 512)
    kf = oracle.frame(kc, dc, BOUNDS)
    free = np.full(len(kc), -1, np.int32)
    a, na = oracle.match_project_sim3(kf, _sim3(T, 1.0), mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 4, free)
    b, nb = oracle.match_project_sim3(kf, _sim3(T, 2.0), mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 4, free)
    assert na > 200 and abs(na - nb) <= 3 and (a == b).mean() > 0.99
    fa, da = oracle.fuse_search_sim3(kf, _sim3(T, 1.0), mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 3.0)
    ok = fa >= 0
    assert ok.sum() > 200 and (fa[ok] == mp["src"][ok]).mean() > 0.95 and (da[ok] <= 0.5).all()
    assert (fa[mp["valid"] == 0] == -1).all()


# ------------------------------------------------------------------ GPU parity through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("n_kf,th,orb,ori", [(2000, 10.0, 1.0, True), (2000, 3.0, 0.64, False), (40, 10.0, 1.0, True)])
def test_match_project_keyframe(hip, oracle, synth, n_kf, th, orb, ori):
    kc, dc = make_frame(2000, 521)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    mp = _map_points_for(kc, dc, T, K, n_kf, 522 + n_kf)
    rng = np.random.default_rng(523)
    occ = (rng.uniform(size=len(kc)) < 0.2).astype(np.uint8)
    ang = (kc["angle"][mp["src"]] + rng.normal(0, 3, n_kf)).astype(np.float32) % np.float32(360)
    hip.frame_set(6, kc, dc, BOUNDS)
    g, gn = hip.match_project_keyframe(6, len(kc), mp["valid"], mp["Xw"], mp["mind"], mp["maxd"], mp["desc"], ang, occ, T, K, th, orb, ori)
    e, en = oracle.match_project_keyframe(oracle.frame(kc, dc, BOUNDS), mp["valid"], mp["Xw"], mp["mind"], mp["maxd"], mp["desc"], ang, occ,
                                          T, K, th, orb, ori)
    np.testing.assert_array_equal(g, e)
    assert gn == en
    if n_kf > 1000:
        assert gn > 200


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 0.37, 2.5])
def test_sim3_projection_and_fuse(hip, oracle, synth, scale):
    kc, dc = make_frame(2000, 531)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    S = _sim3(T, scale)
    mp = _map_points_for(kc, dc, T, K, 3000, 532)
    rng = np.random.default_rng(533)
    matched = np.full(len(kc), -1, np.int32)
    matched[rng.uniform(size=len(kc)) < 0.15] = -2                    # vpMatched already holds something there
    hip.frame_set(7, kc, dc, BOUNDS)
    kf = oracle.frame(kc, dc, BOUNDS)
    g, gn = hip.match_project_sim3(7, S, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 4, matched)
    e, en = oracle.match_project_sim3(kf, S, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 4, matched)
    np.testing.assert_array_equal(g, e)
    assert gn == en and gn > 300
    assert (g[matched == -2] == -2).all()
    gi, gd = hip.fuse_search_sim3(7, S, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 3.0)
    ei, ed = oracle.fuse_search_sim3(kf, S, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 3.0)
    np.testing.assert_array_equal(gi, ei)
    np.testing.assert_array_equal(gd, ed)
    assert (gi >= 0).sum() > 300


@pytest.mark.gpu
def test_match_sim3(hip, oracle, synth):
    """two keyframes of the same scene whose maps differ by a similarity (s12, R12, t12): mutual matches recovered"""
    rng = np.random.default_rng(541)
    K = np.array(synth.KITTI_K, np.float32)
    n = 1500
    k1, d1 = make_frame(n, 542)
    T1 = pose_T()
    depth = rng.uniform(4, 50, n)
    X1 = backproject(T1, K, np.stack([k1["x"], k1["y"]], 1), depth)          # map 1: points of KF1 in world frame 1
    # camera 2 = camera 1 moved a little; its own map lives in a world frame that is a similarity away
    T12 = pose_T((0.004, -0.01, 0.002), (0.15, -0.02, 0.05))                 # X_c1 = R12' X_c2 + t12' (metric)
    s12 = np.float32(1.3)
    R12, t12 = T12[:3, :3].copy(), T12[:3, 3].copy()
    Xc1 = X1.astype(np.float64) @ T1[:3, :3].astype(np.float64).T + T1[:3, 3]
    Xc2 = (Xc1 - t12.astype(np.float64)) @ R12.astype(np.float64) / float(s12)   # X_c2 = (1/s) R12^T (X_c1 - t12)
    vis = Xc2[:, 2] > 0.5
    uv2 = np.stack([K[0] * Xc2[:, 0] / Xc2[:, 2] + K[2], K[1] * Xc2[:, 1] / Xc2[:, 2] + K[3]], 1)
    vis &= (uv2[:, 0] > 20) & (uv2[:, 0] < 1220) & (uv2[:, 1] > 20) & (uv2[:, 1] < 355)
    perm = rng.permutation(n)
    k2 = k1[perm].copy()
    k2["x"] = (uv2[perm, 0] + rng.normal(0, 0.3, n)).astype(np.float32)
    k2["y"] = (uv2[perm, 1] + rng.normal(0, 0.3, n)).astype(np.float32)
    bad = ~vis[perm]
    k2["x"][bad] = rng.uniform(20, 1220, bad.sum()).astype(np.float32)
    k2["y"][bad] = rng.uniform(20, 355, bad.sum()).astype(np.float32)
    d2 = perturbed_descriptors(d1[perm], 0.03, 543)
    T2 = pose_T((0.02, 0.01, -0.01), (-0.3, 0.1, 0.2))                       # arbitrary world frame 2
    X2 = ((Xc2[perm] - T2[:3, 3].astype(np.float64)) @ T2[:3, :3].astype(np.float64)).astype(np.float32)
    has1 = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    has2 = ((rng.uniform(size=n) < 0.9) & ~bad).astype(np.uint8)
    # PredictScale uses the distance in the OTHER camera's frame: make that distance agree with the keypoint's octave
    dist_in_2 = np.linalg.norm(Xc2, axis=1) * rng.uniform(0.95, 1.05, n)          # map-1 points seen from camera 2
    dist_in_1 = np.linalg.norm(Xc1[perm], axis=1) * rng.uniform(0.95, 1.05, n)    # map-2 points seen from camera 1
    maxd1 = (dist_in_2 * SCALES[k1["octave"]]).astype(np.float32); mind1 = (maxd1 / np.float32(SCALES[7])).astype(np.float32)
    maxd2 = (dist_in_1 * SCALES[k2["octave"]]).astype(np.float32); mind2 = (maxd2 / np.float32(SCALES[7])).astype(np.float32)
    hip.frame_set(2, k1, d1, BOUNDS)
    hip.frame_set(3, k2, d2, BOUNDS)
    args = (has1, has2, X1, X2, mind1, maxd1, mind2, maxd2, d1, d2, T1, T2, float(s12), R12, t12, K, 7.5)
    g, gn = hip.match_sim3(2, 3, n, *args)
    e, en = oracle.match_sim3(oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS), *args)
    np.testing.assert_array_equal(g, e)
    assert gn == en and gn > 0.3 * n
    ok = g >= 0
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    assert (g[ok] == inv[np.nonzero(ok)[0]]).mean() > 0.95


@pytest.mark.gpu
def test_degenerate_inputs_do_not_crash(hip, synth):
    """empty keyframes / empty point lists / frames without keypoints return cleanly through every newer entry point"""
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    kc, dc = make_frame(50, 601)
    empty_k, empty_d = kc[:0], dc[:0]
    hip.frame_set(6, kc, dc, BOUNDS)
    hip.frame_set(7, empty_k, empty_d, BOUNDS)
    z3, z1, zd = np.zeros((0, 3), np.float32), np.zeros(0, np.float32), np.zeros((0, 128), np.float32)
    zu = np.zeros(0, np.uint8)
    # no candidate points
    m, n = hip.match_project_keyframe(6, len(kc), zu, z3, z1, z1, zd, z1, np.zeros(len(kc), np.uint8), T, K, 10.0, 1.0, True)
    assert n == 0 and (m == -1).all()
    mk, n = hip.match_project_sim3(6, T, zu, z3, z3, z1, z1, zd, K, 4, np.full(len(kc), -1, np.int32))
    assert n == 0 and (mk == -1).all()
    bi, bd = hip.fuse_search_sim3(6, T, zu, z3, z3, z1, z1, zd, K, 3.0)
    assert len(bi) == 0
    bi, bd = hip.fuse_search(6, zu, z3, z3, z1, z1, zd, T, K, 3.0)
    assert len(bi) == 0
    # keyframe without keypoints
    mp = _map_points_for(kc, dc, T, K, 30, 602)
    bi, bd = hip.fuse_search(7, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], T, K, 3.0)
    assert (bi == -1).all()
    mk, n = hip.match_project_sim3(7, T, mp["valid"], mp["Xw"], mp["normal"], mp["mind"], mp["maxd"], mp["desc"], K, 4, np.zeros(0, np.int32))
    assert n == 0
    m12, n = hip.match_sim3(6, 7, len(kc), np.ones(len(kc), np.uint8), zu, np.zeros((len(kc), 3), np.float32), z3,
                            np.ones(len(kc), np.float32), np.ones(len(kc), np.float32), z1, z1, dc, zd, T, T, 1.0, np.eye(3, dtype=np.float32),
                            np.zeros(3, np.float32), K, 7.5)
    assert n == 0 and (m12 == -1).all()
    # points on the camera plane (z == 0) and behind it are gated, not propagated as NaN
    X0 = np.zeros((4, 3), np.float32)
    X0[:, 2] = [0.0, -5.0, 0.0, 1e-30]
    Xw0 = (X0 - T[:3, 3]) @ T[:3, :3]                       # camera-frame coordinates X0
    ones = np.ones(4, np.float32)
    m, n = hip.match_project_keyframe(6, len(kc), np.ones(4, np.uint8), Xw0.astype(np.float32), ones * 0.1, ones * 100, dc[:4], ones,
                                      np.zeros(len(kc), np.uint8), T, K, 10.0, 1.0, False)
    assert n == (m >= 0).sum()
