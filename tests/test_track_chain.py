"""Fused tracking chains (asd_track_motion_model, asd_track_local_map): the numeric bodies of Tracking::TrackWithMotionModel
(Tracking.cc:664-723) and Tracking::TrackLocalMap (:725-736, :803-851) as one submission each.  They must return exactly what
the separate matcher + PoseOptimization calls return (same kernels, same edge order), and the oracle's matcher + the
reference-g2o-pinned oracle optimiser within the optimiser's tolerance."""
import numpy as np
import pytest

from tests.test_matcher import BOUNDS, SCALES, _m1_case, backproject, make_frame, perturbed_descriptors, pose_T

POSE_TOL = 1e-8


def _pose7(T):
    """Converter::toSE3Quat of a row-major Tcw (f32): unit quaternion + translation as f64"""
    R = T[:3, :3].astype(np.float64)
    t = T[:3, 3].astype(np.float64)
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    return np.concatenate([q / np.linalg.norm(q), t])


def _separate_m1(hip, n_cur, kc, has, Xw, mp_desc, T, K, th, ori, pose0, obs=None):
    m, nm = hip.match_project_frame(0, 1, n_cur, has, Xw, mp_desc, T, K, th, ori, obs_positive=obs)
    j = np.nonzero(m >= 0)[0]
    inv_sigma2 = hip.scale_tables()["inv_sigma2"].astype(np.float64)      # mvInvLevelSigma2 is a float table in the reference
    outl = np.zeros(n_cur, np.uint8)
    pose, ninl = pose0.copy(), 0
    if len(j) >= 3:
        obsv = np.stack([kc["x"][j], kc["y"][j]], 1).astype(np.float64)
        pose, o, ninl = hip.pose_optimize(pose0, Xw[m[j]].astype(np.float64), obsv, inv_sigma2[kc["octave"][j]], K.astype(np.float64))
        outl[j] = o
    return m, nm, pose, outl, ninl


@pytest.mark.gpu
@pytest.mark.parametrize("n,th,ori", [(2000, 15.0, True), (600, 30.0, False), (40, 15.0, True)])
def test_track_motion_model_equals_separate_calls(hip, oracle, synth, n, th, ori):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 400 + n)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    Tg = pose_T(rv=(0.012, -0.018, 0.006), t=(0.12, -0.04, 0.33))        # the motion-model guess the solver starts from
    pose0 = _pose7(Tg)
    exp = _separate_m1(hip, n, kc, has, Xw, mp_desc, T, K, th, ori, pose0)
    got = hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, th, pose0, ori)
    np.testing.assert_array_equal(got[0], exp[0])
    assert got[1] == exp[1] and got[4] == exp[4]
    np.testing.assert_array_equal(got[2], exp[2])                       # same kernels, same edge order: the same bits
    np.testing.assert_array_equal(got[3], exp[3])
    # through the descriptor bank
    hip.bank_put(100, mp_desc)
    gb = hip.track_motion_model(0, 1, n, has, Xw, np.arange(100, 100 + n, dtype=np.int32), T, K, th, pose0, ori)
    for a, b in zip(gb, got):
        np.testing.assert_array_equal(a, b)
    # split in two (asd_track_async / asd_track_finish): inputs may be clobbered once the call has returned, the next frame's
    # frame_set on another slot may run in between
    has2, Xw2, rows2, T2, K2, p2 = has.copy(), Xw.copy(), np.arange(100, 100 + n, dtype=np.int32), T.copy(), K.copy(), pose0.copy()
    assert hip.track_motion_model(0, 1, n, has2, Xw2, rows2, T2, K2, th, p2, ori, split=True) is None
    has2[:] = 0; Xw2[:] = 7; rows2[:] = 0; T2[:] = 0; K2[:] = 1; p2[:] = 9
    hip.frame_set(2, kl, dl, BOUNDS)
    gs = hip.track_finish()
    for a, b in zip(gs, got):
        np.testing.assert_array_equal(a, b)
    # the oracle: matcher bit-exact, optimiser (pinned to the reference's g2o) within its tolerance
    om, onm = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw, mp_desc, T, K, th, ori)
    np.testing.assert_array_equal(got[0], om)
    j = np.nonzero(om >= 0)[0]
    if len(j) >= 3:
        inv_sigma2 = hip.scale_tables()["inv_sigma2"].astype(np.float64)
        op, oo, oi = oracle.pose_optimize(pose0, Xw[om[j]].astype(np.float64), np.stack([kc["x"][j], kc["y"][j]], 1).astype(np.float64),
                                          inv_sigma2[kc["octave"][j]], K.astype(np.float64))
        assert np.abs(got[2] - op).max() <= POSE_TOL and got[4] == oi
        np.testing.assert_array_equal(got[3][j], oo)
    if n >= 600:
        assert got[4] > 0.4 * has.sum()


@pytest.mark.gpu
def test_track_motion_model_projection_on_device_edge_cases(hip, synth):
    """asd_track_motion_model makes the projection loop (ORBmatcher.cc:1343-1368) on the device (k_project_queries); the separate
    asd_match_project_frame call makes it on the host.  Points behind the camera, at depth zero, outside the image on every side,
    exactly on the image border, without a map point, and non-finite positions must fall the same way in both."""
    n = 1200
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 4242)
    rng = np.random.default_rng(7)
    Xw = Xw.copy(); has = has.copy()
    Xw[0:60, 2] *= -1                                  # behind the camera
    Xw[60:70, 2] = -T[2, 3] / max(abs(T[2, 2]), 1e-3)  # depth ~ 0
    Xw[70:130, 0] += rng.choice([-400.0, 400.0], 60).astype(np.float32)   # far outside left / right
    Xw[130:190, 1] += rng.choice([-150.0, 150.0], 60).astype(np.float32)  # above / below
    Xw[190:196] = np.float32(np.inf); Xw[196:200] = np.float32(np.nan)
    has[200:260] = 0
    # a point that projects exactly onto the right image border (u == max_x is inside: the test is u > max_x)
    Xw[260] = backproject(T, K, np.array([[BOUNDS[1], 100.0]], np.float32), np.array([12.0]))[0]
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    pose0 = _pose7(pose_T(rv=(0.01, -0.015, 0.004), t=(0.1, -0.03, 0.3)))
    for th in (15.0, 30.0):
        exp = _separate_m1(hip, n, kc, has, Xw, mp_desc, T, K, th, True, pose0)
        got = hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, th, pose0, True)
        for a, b in zip(got, exp):
            np.testing.assert_array_equal(a, b)
    assert got[1] > 300


@pytest.mark.gpu
def test_track_motion_model_few_matches_and_host_replay(pkg, synth, monkeypatch):
    """< 3 correspondences leave the pose alone (Optimizer.cc:323-324); a context with the host replay runs the same chain
    through the separate entry points and returns the same results"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, 700, 431)
    pose0 = _pose7(pose_T())
    monkeypatch.setenv("ASD_MATCH_REPLAY", "host")
    H = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376)
    monkeypatch.delenv("ASD_MATCH_REPLAY")
    D = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376)
    try:
        res = []
        for dev in (H, D):
            dev.frame_set(0, kc, dc, BOUNDS)
            dev.frame_set(1, kl, dl, BOUNDS)
            res.append(dev.track_motion_model(0, 1, 700, has, Xw, mp_desc, T, K, 15.0, pose0, True))
            two = has.copy(); two[2:] = 0
            m, nm, pose, outl, ninl = dev.track_motion_model(0, 1, 700, two, Xw, mp_desc, T, K, 15.0, pose0, True)
            assert nm <= 2 and ninl == 0 and not outl.any()
            np.testing.assert_array_equal(pose, pose0)
        for a, b in zip(*res):
            np.testing.assert_array_equal(a, b)
    finally:
        H.close()
        D.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_mp,th", [(4000, 1.0), (1500, 5.0)])
def test_track_local_map_equals_separate_calls(hip, oracle, synth, n_mp, th):
    kc, dc = make_frame(2000, 190)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(191 + n_mp)
    src = rng.integers(0, 2000, n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
    depth = rng.uniform(3, 60, n_mp)
    mp_Xw = backproject(T, K, uv, depth)
    hip.frame_set(0, kc, dc, BOUNDS)
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    normal = mp_Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(normal, axis=1)
    normal = (normal / dist[:, None]).astype(np.float32)
    maxd = (dist * SCALES[kc["octave"][src]]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    in_view, proj, level, vc = hip.frustum(0, mp_Xw, normal, mind, maxd, T, K)
    desc = perturbed_descriptors(dc[src], 0.05, 192)
    occupied = (rng.uniform(size=2000) < 0.3).astype(np.uint8)          # keypoints holding a map point from the first stage
    cur_Xw = backproject(T, K, np.stack([kc["x"], kc["y"]], 1) + rng.uniform(-0.7, 0.7, (2000, 2)).astype(np.float32),
                         rng.uniform(4, 50, 2000))
    pose0 = _pose7(pose_T(rv=(0.011, -0.021, 0.004), t=(0.09, -0.06, 0.31)))
    # separate calls
    m, nm = hip.match_project_points(0, 2000, in_view, proj, level, vc, desc, occupied, th, 0.8)
    sel = np.nonzero((occupied > 0) | (m >= 0))[0]
    X = np.where((occupied[sel] > 0)[:, None], cur_Xw[sel], mp_Xw[np.maximum(m[sel], 0)]).astype(np.float64)
    inv_sigma2 = hip.scale_tables()["inv_sigma2"].astype(np.float64)
    pe, oe, ie = hip.pose_optimize(pose0, X, np.stack([kc["x"][sel], kc["y"][sel]], 1).astype(np.float64), inv_sigma2[kc["octave"][sel]],
                                   K.astype(np.float64))
    outl = np.zeros(2000, np.uint8); outl[sel] = oe
    got = hip.track_local_map(0, 2000, in_view, proj, level, vc, desc, mp_Xw, occupied, cur_Xw, th, 0.8, K, pose0)
    np.testing.assert_array_equal(got[0], m)
    assert got[1] == nm and got[4] == ie
    np.testing.assert_array_equal(got[2], pe)
    np.testing.assert_array_equal(got[3], outl)
    assert ie > 0.5 * len(sel)
    hip.bank_put(9000, desc)
    gb = hip.track_local_map(0, 2000, in_view, proj, level, vc, np.arange(9000, 9000 + n_mp, dtype=np.int32), mp_Xw, occupied, cur_Xw, th, 0.8,
                             K, pose0)
    for a, b in zip(gb, got):
        np.testing.assert_array_equal(a, b)
    # oracle matcher on the same inputs
    om, onm = oracle.match_project_points(oracle.frame(kc, dc, BOUNDS), in_view, proj, level, vc, desc, occupied, th, 0.8)
    np.testing.assert_array_equal(got[0], om)


@pytest.mark.gpu
def test_predict_scale_thresholds_equal_logf(pkg):
    """MapPoint::PredictScale (MapPoint.cc:438-453) on the device is a comparison of the distance ratio with thresholds found
    with the host's logf at asd_ctx_create: EVERY float ratio in [0.2, 40] (all levels of a 1.2 pyramid and far beyond both
    clamps) gives the level ceil(logf(r) / logf(scaleFactor)) gives -- also for other pyramids."""
    for scale, levels in ((1.2, 8), (1.5, 5), (1.1, 12)):
        h = pkg.AsdHip(n_features=500, scale_factor=scale, n_levels=levels, max_width=640, max_height=240, max_patches=1000)
        try:
            bad, n = h.debug_level_sweep(0.2, 40.0)
            assert bad == 0 and n > 60_000_000
        finally:
            h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_mp,th", [(4000, 1.0), (900, 3.0)])
def test_track_local_points_equals_frustum_plus_chain(hip, oracle, synth, n_mp, th):
    """asd_track_local_points (frustum test, level prediction and search windows on the device) against asd_frustum on the
    host followed by asd_track_local_map: identical matches, pose bits and outlier flags -- i.e. the device wrote exactly the
    queries the host would have written, for points in front of / behind the camera, outside the image, too near / far, with
    a wrong viewing angle, and at every pyramid level"""
    kc, dc = make_frame(2000, 290)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(291 + n_mp)
    src = rng.integers(0, 2000, n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
    uv[: n_mp // 12] += 2500                                   # outside the image
    depth = rng.uniform(3, 60, n_mp)
    depth[n_mp // 12: n_mp // 8] *= -1                         # behind the camera
    Xw = backproject(T, K, uv, depth)
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    normal = Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(normal, axis=1)
    normal = normal / dist[:, None] + rng.normal(0, 0.45, normal.shape)      # some beyond the 60 degree viewing cone
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    maxd = (dist * SCALES[kc["octave"][src]] * rng.uniform(0.7, 1.4, n_mp)).astype(np.float32)   # all levels, some out of range
    mind = (maxd / np.float32(SCALES[7]) * rng.uniform(0.8, 1.3, n_mp)).astype(np.float32)
    hip.frame_set(0, kc, dc, BOUNDS)
    desc = perturbed_descriptors(dc[src], 0.05, 292)
    occupied = (rng.uniform(size=2000) < 0.3).astype(np.uint8)
    cur_Xw = backproject(T, K, np.stack([kc["x"], kc["y"]], 1) + rng.uniform(-0.7, 0.7, (2000, 2)).astype(np.float32), rng.uniform(4, 50, 2000))
    pose0 = _pose7(pose_T(rv=(0.011, -0.021, 0.004), t=(0.09, -0.06, 0.31)))
    in_view, proj, level, vc = hip.frustum(0, Xw, normal, mind, maxd, T, K)
    oin, oproj, olevel, ovc = oracle.frustum(oracle.frame(kc, dc, BOUNDS), Xw, normal, mind, maxd, T, K)
    np.testing.assert_array_equal(in_view, oin)
    np.testing.assert_array_equal(level, olevel)
    assert 0.2 * n_mp < in_view.sum() < 0.9 * n_mp and len(set(level[in_view > 0])) >= 6
    exp = hip.track_local_map(0, 2000, in_view, proj, level, vc, desc, Xw, occupied, cur_Xw, th, 0.8, K, pose0)
    got = hip.track_local_points(0, 2000, Xw, normal, mind, maxd, desc, T, K, occupied, cur_Xw, th, 0.8, pose0)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)
    hip.bank_put(12000, desc)
    gb = hip.track_local_points(0, 2000, Xw, normal, mind, maxd, np.arange(12000, 12000 + n_mp, dtype=np.int32), T, K, occupied, cur_Xw, th, 0.8, pose0)
    for a, b in zip(gb, exp):
        np.testing.assert_array_equal(a, b)
    assert got[1] > 0 and got[4] > 100
    # the map points by ROW of the descriptor + attribute banks (asd_track_local_points_rows): same attributes, same bits
    hip.mpbank_put(12000, Xw, normal, mind, maxd)
    gr = hip.track_local_points_rows(0, 2000, np.arange(12000, 12000 + n_mp, dtype=np.int32), T, K, occupied, cur_Xw, th, 0.8, pose0)
    for a, b in zip(gr, exp):
        np.testing.assert_array_equal(a, b)
    sub = np.nonzero(rng.uniform(size=n_mp) < 0.6)[0].astype(np.int32)          # a selection of the bank's points, as Tracking names them per frame
    e2 = hip.track_local_points(0, 2000, Xw[sub], normal[sub], mind[sub], maxd[sub], 12000 + sub, T, K, occupied, cur_Xw, th, 0.8, pose0)
    assert hip.track_local_points_rows(0, 2000, 12000 + sub, T, K, occupied, cur_Xw, th, 0.8, pose0, split=True) is None
    g2 = hip.track_finish()
    for a, b in zip(g2, e2):
        np.testing.assert_array_equal(a, b)
    # split in two: every input is consumed when the call returns
    cp = [a.copy() for a in (Xw, normal, mind, maxd, occupied, cur_Xw, pose0)]
    assert hip.track_local_points(0, 2000, cp[0], cp[1], cp[2], cp[3], desc, T, K, cp[4], cp[5], th, 0.8, cp[6], split=True) is None
    for a in cp:
        a[:] = 3
    k2, d2 = make_frame(500, 77)
    hip.frame_set(1, k2, d2, BOUNDS)                 # the next frame's grid, behind the chain on the same stream
    gs = hip.track_finish()
    for a, b in zip(gs, exp):
        np.testing.assert_array_equal(a, b)
    # same through asd_track_local_map
    assert hip.track_local_map(0, 2000, in_view, proj, level, vc, desc, Xw, occupied, cur_Xw, th, 0.8, K, pose0, split=True) is None
    gs = hip.track_finish()
    for a, b in zip(gs, exp):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_split_phase_rules(hip, synth):
    """asd_track_async arms exactly the next asd_track_* call; while that call is outstanding the matcher and the pose solver are
    refused (their buffers are in use), asd_track_finish without an outstanding call is an error, and a failing armed call leaves
    nothing outstanding."""
    n = 300
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 911)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    pose0 = _pose7(pose_T())
    with pytest.raises(Exception):
        hip.track_finish()
    ref = hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, 15.0, pose0)
    assert hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, 15.0, pose0, split=True) is None
    with pytest.raises(Exception):
        hip.match_project_frame(0, 1, n, has, Xw, mp_desc, T, K, 15.0, True)
    with pytest.raises(Exception):
        hip.pose_optimize(pose0, Xw[:50].astype(np.float64), np.zeros((50, 2)), np.ones(50), K.astype(np.float64))
    with pytest.raises(Exception):
        hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, 15.0, pose0)
    got = hip.track_finish()
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    # an armed call that is refused (bad slot) leaves nothing outstanding and does not stay armed
    with pytest.raises(Exception):
        hip.track_motion_model(0, 99, n, has, Xw, mp_desc, T, K, 15.0, pose0, split=True)
    with pytest.raises(Exception):
        hip.track_finish()
    again = hip.track_motion_model(0, 1, n, has, Xw, mp_desc, T, K, 15.0, pose0)
    for a, b in zip(again, ref):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_deferred_stage_never_searches_again(pkg, synth):
    """A candidate-buffer overflow used to be handled at asd_track_finish by searching again -- over bank rows and frame slots the
    caller may have rewritten in between (include/asd_slam.h allows asd_bank_put* / asd_frame_set there).  A deferred stage now sizes
    the candidate buffers for the worst case before it enqueues (a query's list holds at most every keypoint of the frame), so its
    completion never retries.  Fresh contexts (candidate buffers at their initial 262144 entries), a search radius whose lists
    exceed that, and the bank rows overwritten between the armed call and asd_track_finish: same bits as the synchronous call."""
    n, th = 2000, 170.0
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 4242)
    pose0 = _pose7(pose_T(rv=(0.012, -0.018, 0.006), t=(0.12, -0.04, 0.33)))
    rows = np.arange(n, dtype=np.int32)
    # the lists really overflow the initial buffers: count the window candidates of every query (ORBmatcher.cc:1376-1383)
    Xc = Xw.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
    u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
    v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    total = 0
    for i in np.nonzero(has)[0]:
        r = th * SCALES[kl["octave"][i]]
        lv = (kc["octave"] >= kl["octave"][i] - 1) & (kc["octave"] <= kl["octave"][i] + 1)
        total += int((lv & (np.abs(kc["x"] - u[i]) < r) & (np.abs(kc["y"] - v[i]) < r)).sum())
    assert total > 300000, total

    def ctx():
        h = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
        h.frame_set(0, kc, dc, BOUNDS)
        h.frame_set(1, kl, dl, BOUNDS)
        h.bank_put(0, mp_desc)
        return h
    a = ctx()
    try:
        exp = a.track_motion_model(0, 1, n, has, Xw, rows, T, K, th, pose0, True)   # synchronous: grows and searches again, nothing changes meanwhile
    finally:
        a.close()
    assert exp[1] > 100
    b = ctx()
    try:
        assert b.track_motion_model(0, 1, n, has, Xw, rows, T, K, th, pose0, True, split=True) is None
        b.bank_put(0, np.ascontiguousarray(mp_desc[::-1]))    # the rows the pending stage matched against are rewritten ...
        b.frame_set(2, kl, dl, BOUNDS)                         # ... and another slot is filled, as the next frame's construction does
        got = b.track_finish()
    finally:
        b.close()
    for x, y in zip(got, exp):
        np.testing.assert_array_equal(x, y)


def _frame_case(synth, n, seed):
    """a tracked-frame situation: last frame + current frame (as _m1_case), and a local map = the last frame's points plus a displaced
    copy of each (twice the keypoints, like bench.py's stand-in), every last keypoint holding candidate i"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, seed)
    Xw2 = np.concatenate([Xw, Xw + np.float32(0.02)])
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    d = Xw2.astype(np.float64) - Ow
    dist = np.linalg.norm(d, axis=1)
    nrm = (d / dist[:, None]).astype(np.float32)
    lv = np.concatenate([kl["octave"], kl["octave"]])
    maxd = (dist * SCALES[lv]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    return kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd


@pytest.mark.gpu
@pytest.mark.parametrize("n,th,flags", [(2000, 15.0, False), (700, 30.0, True), (2, 15.0, False)])
def test_track_frame_equals_the_two_stage_calls(hip, synth, n, th, flags):
    """asd_track_frame (both stages + what Tracking does between them, one submission) against asd_track_motion_model_bank -> host ->
    asd_track_local_points_bank: the same bits in every output"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd = _frame_case(synth, n, 900 + n)
    n_cur = len(kc)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    base = 300
    hip.bank_put(base, np.concatenate([mp_desc, mp_desc]))
    hip.mpbank_put(base, Xw2, nrm, mind, maxd)
    rows1 = np.arange(base, base + n, dtype=np.int32)
    cand_rows = np.arange(base, base + 2 * n, dtype=np.int32)
    rng = np.random.default_rng(4)
    obs1 = (rng.uniform(size=n) < 0.8).astype(np.uint8) if flags else None
    obs2 = (rng.uniform(size=2 * n) < 0.8).astype(np.uint8) if flags else None
    last_cand = np.arange(n, dtype=np.int32)
    if flags:
        last_cand[::7] = -1            # map points that are not among the candidates
    pose0 = _pose7(pose_T(rv=(0.012, -0.018, 0.006), t=(0.12, -0.04, 0.33)))
    # ---- the two calls with the host in between (bench.py's track_step)
    m1, n1, pose1, outl1, inl1 = hip.track_motion_model(0, 1, n_cur, has, Xw, rows1, T, K, th, pose0, True, obs_positive=obs1)
    keep = (m1 >= 0) & (outl1 == 0)
    T1 = hip.pose7_to_tcw(pose1) if (m1 >= 0).sum() >= 3 else T
    in_frame = np.zeros(2 * n, bool)
    lc = last_cand[m1[m1 >= 0]]      # kept matches and dropped outliers alike: neither map point is searched again (Tracking.cc:705-707, :811-823)
    in_frame[lc[lc >= 0]] = True
    sel = np.nonzero(~in_frame)[0].astype(np.int32)
    occ = keep.astype(np.uint8)
    cur_Xw = Xw[np.maximum(m1, 0)]
    m2, n2, pose2, outl2, inl2 = hip.track_local_points(0, n_cur, Xw2[sel], nrm[sel], mind[sel], maxd[sel], cand_rows[sel], T1, K, occ, cur_Xw, 1.0, 0.8,
                                                        pose1, obs_positive=None if obs2 is None else obs2[sel])
    # ---- one submission
    for split in (False, True):
        r = hip.track_frame(0, 1, n_cur, has, Xw, rows1, last_cand, T, K, th, pose0, cand_rows, 1.0, 0.8, last_obs_positive=obs1, cand_obs_positive=obs2,
                            split=split)
        if split:
            assert r is None
            hip.frame_set(2, kl, dl, BOUNDS)      # the next frame's construction may run meanwhile
            r = hip.track_frame_finish()
        np.testing.assert_array_equal(r["match1"], m1)
        assert (r["n1"], r["n_inl1"]) == (n1, inl1)
        np.testing.assert_array_equal(r["outlier1"], outl1)
        np.testing.assert_array_equal(r["pose1"], pose1)
        exp2 = np.where(m2 >= 0, sel[np.maximum(m2, 0)], -1)     # candidate index instead of index into the compacted list
        np.testing.assert_array_equal(r["match2"], exp2)
        assert (r["n2"], r["n_inl2"]) == (n2, inl2)
        np.testing.assert_array_equal(r["outlier2"], outl2)
        np.testing.assert_array_equal(r["pose"], pose2)
    if n >= 700:
        assert n1 > 0.3 * n and n2 > 0 and inl2 > 0.3 * n


def _active_lists(oframe, in_view, proj, level, vc, th, occ):
    """map points whose search window holds at least one unoccupied keypoint (ORBmatcher.cc:62-84): the lists the replay walks"""
    n = 0
    for q in np.nonzero(in_view)[0]:
        r = (2.5 if vc[q] > 0.998 else 4.0) * (th if th != 1.0 else 1.0) * SCALES[level[q]]
        idx = oframe.features_in_area(np.float32(proj[q, 0]), np.float32(proj[q, 1]), np.float32(r), int(level[q]) - 1, int(level[q]))
        n += bool(len(idx)) and not occ[idx].all()
    return n


def _oracle_frame_composition(oracle, inv_sigma2, kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd, last_cand, th, pose0, obs1, obs2, hip_pose1=None, cand_desc=None, th_local=1.0):
    """One tracked frame as the ORACLE composes it: SearchByProjection(cur, last) (ORBmatcher.cc:1318-1452) -> PoseOptimization
    (Optimizer.cc:239-413) -> what Tracking does between the stages (outliers dropped :695-714 with mnLastFrameSeen stamped :705-707, pose
    hand-over Frame.cc:150-158, SearchLocalPoints' skip marks :811-823) -> isInFrustum (Frame.cc:160-217) -> SearchByProjection(cur, points)
    (:44-122) -> PoseOptimization.  hip_pose1: the product's stage-1 pose handed to the stages behind it (the optimiser is tolerance-checked,
    1e-8; everything behind it is compared bit for bit and must start from the same bits)."""
    n, n_cur = len(kl), len(kc)
    oc, ol = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
    K64 = K.astype(np.float64)
    m1, n1 = oracle.match_project_frame(oc, ol, has, Xw, mp_desc, T, K, th, True, obs_positive=obs1)
    j = np.nonzero(m1 >= 0)[0]
    outl1, pose1, inl1 = np.zeros(n_cur, np.uint8), pose0.copy(), 0
    if len(j) >= 3:
        pose1, o, inl1 = oracle.pose_optimize(pose0, Xw[m1[j]].astype(np.float64), np.stack([kc["x"][j], kc["y"][j]], 1).astype(np.float64),
                                              inv_sigma2[kc["octave"][j]], K64)
        outl1[j] = o
    keep = (m1 >= 0) & (outl1 == 0)
    hand = pose1 if hip_pose1 is None else hip_pose1
    T1 = oracle.pose7_to_tcw(hand) if len(j) >= 3 else T
    in_frame = np.zeros(len(Xw2), bool)
    lc = last_cand[m1[m1 >= 0]]
    in_frame[lc[lc >= 0]] = True
    sel = np.nonzero(~in_frame)[0].astype(np.int32)
    occ = keep.astype(np.uint8)
    in_view, proj, level, vc = oracle.frustum(oc, Xw2[sel], nrm[sel], mind[sel], maxd[sel], T1, K)
    d2 = (np.concatenate([mp_desc, mp_desc]) if cand_desc is None else cand_desc)[sel]
    m2s, n2 = oracle.match_project_points(oc, in_view, proj, level, vc, d2, occ, th_local, 0.8, obs_positive=None if obs2 is None else obs2[sel])
    n_active = _active_lists(oc, in_view, proj, level, vc, th_local, occ)
    m2 = np.where(m2s >= 0, sel[np.maximum(m2s, 0)], -1)          # candidate indices
    jj = np.nonzero(keep | (m2 >= 0))[0]
    outl2, pose2, inl2 = np.zeros(n_cur, np.uint8), hand.copy(), 0
    if len(jj) >= 3:
        X = np.where(keep[jj][:, None], Xw[np.maximum(m1[jj], 0)], Xw2[np.maximum(m2[jj], 0)]).astype(np.float64)
        pose2, o, inl2 = oracle.pose_optimize(hand, X, np.stack([kc["x"][jj], kc["y"][jj]], 1).astype(np.float64), inv_sigma2[kc["octave"][jj]], K64)
        outl2[jj] = o
    return dict(m1=m1, n1=n1, pose1=pose1, outl1=outl1, inl1=inl1, sel=sel, m2=m2, n2=n2, pose2=pose2, outl2=outl2, inl2=inl2, keep=keep, n_active=n_active)


@pytest.mark.gpu
@pytest.mark.parametrize("n,flags", [(2000, False), (2000, True), (500, True)])
def test_tracked_frame_against_the_oracle_composition(hip, oracle, synth, n, flags):
    """The kernels the bench times, held to the oracle DIRECTLY at the bench's size (2000 keypoints, 4000 local-map candidates, Observations()
    flags off and on): asd_track_frame (k_resolve_pose<0,4> / <1,8>, k_window_search<true>, the bank form of k_frustum_queries, asd_between_body)
    and the two-call form the headline runs (asd_track_motion_model_bank -> host -> asd_track_local_points_bank).  Match ids, match counts and
    outlier flags bit-exact; poses within the optimiser's 1e-8."""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd = _frame_case(synth, n, 2300 + n + int(flags))
    n_cur = len(kc)
    # a tenth of the keypoints a few pixels off their map point's projection: matched by the motion-model stage, marked as outliers by its
    # PoseOptimization, dropped between the stages -- keypoint free again, map point NOT searched again (Tracking.cc:695-714)
    rng = np.random.default_rng(15)
    off = rng.choice(n_cur, n_cur // 10, replace=False)
    kc = kc.copy()
    kc["x"][off] += rng.choice([-1.0, 1.0], len(off)).astype(np.float32) * rng.uniform(3.5, 6.0, len(off)).astype(np.float32)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    base = 500
    hip.bank_put(base, np.concatenate([mp_desc, mp_desc]))
    hip.mpbank_put(base, Xw2, nrm, mind, maxd)
    rows1 = np.arange(base, base + n, dtype=np.int32)
    cand_rows = np.arange(base, base + 2 * n, dtype=np.int32)
    rng = np.random.default_rng(14)
    obs1 = (rng.uniform(size=n) < 0.8).astype(np.uint8) if flags else None
    obs2 = (rng.uniform(size=2 * n) < 0.8).astype(np.uint8) if flags else None
    last_cand = np.arange(n, dtype=np.int32)
    if flags:
        last_cand[::9] = -1
    pose0 = _pose7(pose_T(rv=(0.012, -0.018, 0.006), t=(0.12, -0.04, 0.33)))
    inv_sigma2 = hip.scale_tables()["inv_sigma2"].astype(np.float64)
    # ---- one submission
    r = hip.track_frame(0, 1, n_cur, has, Xw, rows1, last_cand, T, K, 15.0, pose0, cand_rows, 1.0, 0.8, last_obs_positive=obs1, cand_obs_positive=obs2)
    o = _oracle_frame_composition(oracle, inv_sigma2, kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd, last_cand, 15.0, pose0, obs1, obs2,
                                  hip_pose1=r["pose1"])
    np.testing.assert_array_equal(r["match1"], o["m1"])
    assert (r["n1"], r["n_inl1"]) == (o["n1"], o["inl1"])
    np.testing.assert_array_equal(r["outlier1"], o["outl1"])
    assert np.abs(r["pose1"] - o["pose1"]).max() <= POSE_TOL
    np.testing.assert_array_equal(r["match2"], o["m2"])
    assert (r["n2"], r["n_inl2"]) == (o["n2"], o["inl2"])
    np.testing.assert_array_equal(r["outlier2"], o["outl2"])
    assert np.abs(r["pose"] - o["pose2"]).max() <= POSE_TOL
    # ---- the two calls with the host between them (what bench.py's `value` runs): same oracle expectation
    m1, n1, pose1, outl1, inl1 = hip.track_motion_model(0, 1, n_cur, has, Xw, rows1, T, K, 15.0, pose0, True, obs_positive=obs1)
    np.testing.assert_array_equal(m1, o["m1"])
    np.testing.assert_array_equal(outl1, o["outl1"])
    np.testing.assert_array_equal(pose1, r["pose1"])
    T1 = hip.pose7_to_tcw(pose1) if (m1 >= 0).sum() >= 3 else T
    sel = o["sel"]
    m2, n2, pose2, outl2, inl2 = hip.track_local_points(0, n_cur, Xw2[sel], nrm[sel], mind[sel], maxd[sel], cand_rows[sel], T1, K, o["keep"].astype(np.uint8),
                                                        Xw[np.maximum(m1, 0)], 1.0, 0.8, pose1, obs_positive=None if obs2 is None else obs2[sel])
    np.testing.assert_array_equal(np.where(m2 >= 0, sel[np.maximum(m2, 0)], -1), o["m2"])
    assert (n1, inl1, n2, inl2) == (o["n1"], o["inl1"], o["n2"], o["inl2"])
    np.testing.assert_array_equal(outl2, o["outl2"])
    assert np.abs(pose2 - o["pose2"]).max() <= POSE_TOL
    assert o["n1"] > 0.3 * n and o["n2"] > 0 and o["inl2"] > 0.3 * n
    assert o["outl1"].sum() > 0.02 * n            # the outlier rule between the stages is exercised


@pytest.mark.gpu
@pytest.mark.parametrize("copies", [4, 8])
def test_large_local_map_against_the_oracle(hip, oracle, synth, copies):
    """Local maps beyond one chunk of the replay workgroup (Tracking.cc:881-905 collects every point of up to 80 local keyframes): 8000 and
    16000 candidates for a frame of 2000 keypoints, most of them in view with a candidate list -- the replay runs over the map points that
    HAVE a list in chunks of 4096, in index order (resolve2.h).  asd_track_frame and the two-call form against the oracle's composition."""
    n = 2000
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, _, _, _, _ = _frame_case(synth, n, 5100 + copies)
    n_cur = len(kc)
    Xw2 = np.concatenate([Xw + np.float32(0.015 * j) for j in range(copies)])
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    d = Xw2.astype(np.float64) - Ow
    dist = np.linalg.norm(d, axis=1)
    nrm = (d / dist[:, None]).astype(np.float32)
    lv = np.concatenate([kl["octave"]] * copies)
    maxd = (dist * SCALES[lv]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    cand_desc = np.concatenate([perturbed_descriptors(mp_desc, 0.01 * j, 77 + j) if j else mp_desc for j in range(copies)])
    ncand = copies * n
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    base = 700
    hip.bank_put(base, cand_desc)
    hip.mpbank_put(base, Xw2, nrm, mind, maxd)
    rows1 = np.arange(base, base + n, dtype=np.int32)
    cand_rows = np.arange(base, base + ncand, dtype=np.int32)
    rng = np.random.default_rng(3)
    obs2 = (rng.uniform(size=ncand) < 0.85).astype(np.uint8)
    last_cand = np.arange(n, dtype=np.int32)
    pose0 = _pose7(pose_T(rv=(0.012, -0.018, 0.006), t=(0.12, -0.04, 0.33)))
    inv_sigma2 = hip.scale_tables()["inv_sigma2"].astype(np.float64)
    r = hip.track_frame(0, 1, n_cur, has, Xw, rows1, last_cand, T, K, 15.0, pose0, cand_rows, 3.0, 0.8, cand_obs_positive=obs2)
    o = _oracle_frame_composition_th(oracle, inv_sigma2, kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd, last_cand, 15.0, pose0, None, obs2,
                                     hip_pose1=r["pose1"], cand_desc=cand_desc, th_local=3.0)
    np.testing.assert_array_equal(r["match1"], o["m1"])
    np.testing.assert_array_equal(r["match2"], o["m2"])
    assert (r["n1"], r["n_inl1"], r["n2"], r["n_inl2"]) == (o["n1"], o["inl1"], o["n2"], o["inl2"])
    np.testing.assert_array_equal(r["outlier2"], o["outl2"])
    assert np.abs(r["pose"] - o["pose2"]).max() <= POSE_TOL
    if copies == 8:
        assert o["n_active"] > 4096, o["n_active"]      # more lists than one chunk of either replay workgroup takes
    # the two calls with the host between them
    m1, n1, pose1, outl1, inl1 = hip.track_motion_model(0, 1, n_cur, has, Xw, rows1, T, K, 15.0, pose0, True)
    np.testing.assert_array_equal(m1, o["m1"])
    T1 = hip.pose7_to_tcw(pose1) if (m1 >= 0).sum() >= 3 else T
    sel = o["sel"]
    m2, n2, pose2, outl2, inl2 = hip.track_local_points(0, n_cur, Xw2[sel], nrm[sel], mind[sel], maxd[sel], cand_rows[sel], T1, K, o["keep"].astype(np.uint8),
                                                        Xw[np.maximum(m1, 0)], 3.0, 0.8, pose1, obs_positive=obs2[sel])
    np.testing.assert_array_equal(np.where(m2 >= 0, sel[np.maximum(m2, 0)], -1), o["m2"])
    assert (n2, inl2) == (o["n2"], o["inl2"])
    np.testing.assert_array_equal(outl2, o["outl2"])


def _oracle_frame_composition_th(*args, th_local=1.0, **kw):
    return _oracle_frame_composition(*args, **kw, th_local=th_local)


@pytest.mark.gpu
def test_track_frame_refuses_what_it_cannot_chain(hip, synth):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd = _frame_case(synth, 300, 77)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    hip.bank_put(0, np.concatenate([mp_desc, mp_desc]))
    hip.mpbank_put(0, Xw2, nrm, mind, maxd)
    pose0 = _pose7(pose_T())
    with pytest.raises(RuntimeError):        # candidate row beyond the banks
        hip.track_frame(0, 1, 300, has, Xw, np.arange(300, dtype=np.int32), None, T, K, 15.0, pose0, np.arange(10_000_000, 10_000_600, dtype=np.int32), 1.0, 0.8)
    with pytest.raises(RuntimeError):        # more candidates than the compaction in front of the replay takes (64 rounds of 512)
        hip.track_frame(0, 1, 300, has, Xw, np.arange(300, dtype=np.int32), None, T, K, 15.0, pose0, np.zeros(40000, np.int32), 1.0, 0.8)
    big = hip.track_frame(0, 1, 300, has, Xw, np.arange(300, dtype=np.int32), None, T, K, 15.0, pose0, np.tile(np.arange(600, dtype=np.int32), 9)[:5000], 1.0, 0.8)
    assert big["n1"] > 0                     # 5000 candidates (round 4 refused more than 4096)
    r = hip.track_frame(0, 1, 300, has, Xw, np.arange(300, dtype=np.int32), None, T, K, 15.0, pose0, np.arange(600, dtype=np.int32), 1.0, 0.8)
    assert r["n1"] > 0


@pytest.mark.gpu
def test_track_local_points_rows_refuses_bad_rows(hip, synth):
    """asd_track_local_points_rows names its map points by bank row: a row outside the banks is an error of the call, reported before
    anything is enqueued (no stage left outstanding), and the same call with good rows goes through afterwards."""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, Xw2, nrm, mind, maxd = _frame_case(synth, 300, 78)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.bank_put(0, np.concatenate([mp_desc, mp_desc]))
    hip.mpbank_put(0, Xw2, nrm, mind, maxd)
    pose0 = _pose7(pose_T())
    occ = np.zeros(300, np.uint8)
    cur_Xw = np.zeros((300, 3), np.float32)
    rows = np.arange(600, dtype=np.int32)
    bad = rows.copy()
    bad[17] = 50_000_000
    with pytest.raises(RuntimeError):
        hip.track_local_points_rows(0, 300, bad, T, K, occ, cur_Xw, 1.0, 0.8, pose0)
    neg = rows.copy()
    neg[0] = -1
    with pytest.raises(RuntimeError):
        hip.track_local_points_rows(0, 300, neg, T, K, occ, cur_Xw, 1.0, 0.8, pose0, split=True)
    good = hip.track_local_points_rows(0, 300, rows, T, K, occ, cur_Xw, 1.0, 0.8, pose0)
    assert good[1] > 0
