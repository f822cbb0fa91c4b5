"""BASELINE.json configs[4] = the 11 KITTI odometry sequences.  Their three camera classes differ in image size and intrinsics
(the reference's cameraconfig/KITTI/kitti00-02.txt, kitti03.txt, kitti04-12.txt, read by kitti.cc:56-86): 1241x376 is what every
other test runs at; this file runs the HIP path at 1242x375 (sequence 03) and 1226x370 (04-10) -- extraction bit-exact against the
oracle, and the fused tracking chains (TrackWithMotionModel / TrackLocalMap bodies) on frames extracted at that size with that
camera's intrinsics -- and drives two contexts through the read-ahead extractor at the same time (what a stereo rig or two
sequences in one process do)."""
import threading

import numpy as np
import pytest

from tests.test_frontend import _compare_extract
from tests.test_matcher import SCALES
from tests.test_track_chain import POSE_TOL, _pose7

# (width, height, fx, fy, cx, cy): cameraconfig/KITTI/kitti03.txt, kitti04-12.txt
KITTI03 = (1242, 375, (721.5377, 721.5377, 609.5593, 172.854))
KITTI04_12 = (1226, 370, (707.0912, 707.09127, 601.8873, 183.1104))


def _ctx(pkg, synth, w, h):
    c = pkg.AsdHip(n_features=2000, max_width=w, max_height=h, max_patches=4096)
    c.load_weights(synth.asdnet_weights(0))
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("cam", [KITTI03, KITTI04_12], ids=["kitti03_1242x375", "kitti04-12_1226x370"])
def test_extract_bit_exact_at_kitti_camera_sizes(pkg, oracle, synth, cam):
    w, h, _ = cam
    ctx = _ctx(pkg, synth, w, h)
    try:
        kps, _ = _compare_extract(ctx, oracle, synth, synth.scene_frame(3, seed=23, w=w, h=h), 2000)
        assert len(kps) >= 2000
        # the 16-px FAST border and the 19-px patch border at THIS size
        assert kps["x"].min() >= 19 and kps["x"].max() < w and kps["y"].min() >= 19 and kps["y"].max() < h
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cam", [KITTI03, KITTI04_12], ids=["kitti03_1242x375", "kitti04-12_1226x370"])
def test_fused_chains_at_kitti_camera_sizes(pkg, oracle, synth, cam):
    """frame t and t+1 of the synthetic stream at the sequence's size -> extract -> grid -> asd_track_motion_model (projection,
    SearchByProjection(frame, frame), PoseOptimization) -> asd_track_local_points (isInFrustum, SearchByProjection(frame, points),
    PoseOptimization) with the camera's intrinsics: matches bit-exact against the oracle's matchers, poses within the optimiser's
    tolerance, and the fused calls equal to the separate ones."""
    w, h, Kt = cam
    K = np.array(Kt, np.float32)
    bounds = (0.0, float(w), 0.0, float(h))
    ctx = _ctx(pkg, synth, w, h)
    try:
        kl, dl = ctx.extract(synth.scene_frame(4, seed=23, w=w, h=h))
        kl, dl = kl.copy(), dl.copy()
        kc, dc = ctx.extract(synth.scene_frame(5, seed=23, w=w, h=h))
        kc, dc = kc.copy(), dc.copy()
        n_cur, nl = len(kc), len(kl)
        ctx.frame_set(0, kc, dc, bounds)
        ctx.frame_set(1, kl, dl, bounds)
        # the last frame's map points: its keypoints where the stream's drift puts them in frame t+1, at 20 m, identity pose
        z = np.float32(1.003)
        uv = np.stack([(kl["x"] - np.float32(w / 2)) * z + np.float32(w / 2) - 3 * z, (kl["y"] - np.float32(h / 2)) * z + np.float32(h / 2) - np.float32(0.2) * z], 1).astype(np.float32)
        Xw = np.stack([(uv[:, 0] - K[2]) / K[0] * 20.0, (uv[:, 1] - K[3]) / K[1] * 20.0, np.full(nl, 20.0)], 1).astype(np.float32)
        has = np.ones(nl, np.uint8)
        T = np.eye(4, dtype=np.float32)
        pose0 = _pose7(T)
        # ---- motion-model stage
        got = ctx.track_motion_model(0, 1, n_cur, has, Xw, dl, T, K, 15.0, pose0, True)
        m, nm = ctx.match_project_frame(0, 1, n_cur, has, Xw, dl, T, K, 15.0, True)
        om, onm = oracle.match_project_frame(oracle.frame(kc, dc, bounds), oracle.frame(kl, dl, bounds), has, Xw, dl, T, K, 15.0, True)
        np.testing.assert_array_equal(got[0], om)
        np.testing.assert_array_equal(m, om)
        assert got[1] == onm == nm and onm > 0.4 * nl
        inv_sigma2 = ctx.scale_tables()["inv_sigma2"].astype(np.float64)
        j = np.nonzero(om >= 0)[0]
        obs = np.stack([kc["x"][j], kc["y"][j]], 1).astype(np.float64)
        sp, so, si = ctx.pose_optimize(pose0, Xw[om[j]].astype(np.float64), obs, inv_sigma2[kc["octave"][j]], K.astype(np.float64))
        np.testing.assert_array_equal(got[2], sp)                      # fused == separate: the same bits
        assert got[4] == si
        op, oo, oi = oracle.pose_optimize(pose0, Xw[om[j]].astype(np.float64), obs, inv_sigma2[kc["octave"][j]], K.astype(np.float64))
        assert np.abs(got[2] - op).max() <= POSE_TOL and got[4] == oi
        np.testing.assert_array_equal(got[3][j], oo)
        # ---- local-map stage: the same points plus a displaced copy, level ranges from the last frame's octaves
        Xw2 = np.concatenate([Xw, Xw + np.float32(0.02)])
        nrm = (Xw2 / np.linalg.norm(Xw2, axis=1, keepdims=True)).astype(np.float32)
        dist = np.linalg.norm(Xw2, axis=1).astype(np.float32)
        lv = np.concatenate([kl["octave"], kl["octave"]])
        maxd = dist * SCALES[lv]
        mind = maxd / SCALES[7]
        d2 = np.concatenate([dl, dl])
        occ = (om >= 0).astype(np.uint8)
        cur_Xw = Xw[np.maximum(om, 0)]
        g2 = ctx.track_local_points(0, n_cur, Xw2, nrm, mind, maxd, d2, T, K, occ, cur_Xw, 1.0, 0.8, pose0)
        in_view, proj, level, vc = ctx.frustum(0, Xw2, nrm, mind, maxd, T, K)
        oin, oproj, olevel, ovc = oracle.frustum(oracle.frame(kc, dc, bounds), Xw2, nrm, mind, maxd, T, K)
        np.testing.assert_array_equal(in_view, oin)
        np.testing.assert_array_equal(level, olevel)
        om2, on2 = oracle.match_project_points(oracle.frame(kc, dc, bounds), oin, oproj, olevel, ovc, d2, occ, 1.0, 0.8)
        np.testing.assert_array_equal(g2[0], om2)
        assert g2[1] == on2
        jj = np.nonzero((om >= 0) | (om2 >= 0))[0]
        X = np.where((om[jj] >= 0)[:, None], Xw[np.maximum(om[jj], 0)], Xw2[np.maximum(om2[jj], 0)])
        obs2 = np.stack([kc["x"][jj], kc["y"][jj]], 1).astype(np.float64)
        op2, oo2, oi2 = oracle.pose_optimize(pose0, X.astype(np.float64), obs2, inv_sigma2[kc["octave"][jj]], K.astype(np.float64))
        assert np.abs(g2[2] - op2).max() <= POSE_TOL and g2[4] == oi2 and oi2 > 500
        np.testing.assert_array_equal(g2[3][jj], oo2)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_two_contexts_through_the_readahead_extractor_at_once(pkg, synth):
    """Two contexts of one process (the stereo pair of configs[3], or two sequences) keep their read-ahead queues full from two
    host threads at the same time -- their extractor workers and streams share the device -- and every frame comes back exactly
    as the sequential asd_extract returns it, in submission order."""
    sizes = [(752, 480), (1226, 370)]
    frames = [[synth.scene_frame(t, seed=31 + k, w=w, h=h) for t in range(6)] for k, (w, h) in enumerate(sizes)]
    ctxs = [_ctx(pkg, synth, w, h) for (w, h) in sizes]
    try:
        expect = [[tuple(a.copy() for a in c.extract(f)) for f in fr] for c, fr in zip(ctxs, frames)]
        results = [[], []]
        errors = []

        dev = []
        for c, fr in zip(ctxs, frames):   # frames resident in HBM, as the replay keeps them
            hs = []
            for f in fr:
                p = c.device_alloc(f.nbytes)
                c.h2d(p, f)
                hs.append(p)
            dev.append(hs)

        def drive(i):
            try:
                c, fr = ctxs[i], dev[i]
                w, h = sizes[i]
                c.extract_submit(fr[0], w, h, w, device_resident=True)
                c.extract_submit(fr[1], w, h, w, device_resident=True)
                for t in range(len(fr)):
                    k, d = c.extract_wait()
                    results[i].append((k.copy(), d.copy()))
                    if t + 2 < len(fr):
                        c.extract_submit(fr[t + 2], w, h, w, device_resident=True)
            except Exception as e:   # noqa: BLE001 -- reported by the main thread
                errors.append((i, repr(e)))
        th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not errors, errors
        for i in range(2):
            assert len(results[i]) == len(frames[i])
            for (k, d), (ek, ed) in zip(results[i], expect[i]):
                np.testing.assert_array_equal(k, ek)
                np.testing.assert_array_equal(d, ed)
    finally:
        for c in ctxs:
            c.close()
