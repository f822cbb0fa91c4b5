"""Local-mapping triangulation (LocalMapping::CreateNewMapPoints per-match body, LocalMapping.cc:386-519):
oracle known-answer checks on CPU, HIP-vs-oracle parity on the GPU through the C ABI."""
import numpy as np
import pytest

from tests.test_matcher import BOUNDS, SCALES, make_frame, pose_T

K_KITTI = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)


def _project(T, K, X):
    Xc = X.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
    return np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1), Xc[:, 2]


def _two_keyframes(n, seed, noise=0.4, baseline=1.0, wrong_frac=0.15):
    """n 3-D points seen by two keyframes `baseline` apart; keypoints = projections + pixel noise; a fraction of the
    pairs is deliberately mismatched, some points sit at near-zero parallax (very far)"""
    rng = np.random.default_rng(seed)
    T1 = pose_T((0.01, -0.02, 0.005), (0.1, -0.05, 0.3))
    T2 = pose_T((0.012, -0.05, 0.0), (0.1 - baseline, -0.02, 0.1))
    k1, _ = make_frame(n, seed + 1)
    depth = rng.uniform(4, 35, n)
    depth[: n // 10] = rng.uniform(3000, 9000, n // 10)  # parallax below the 0.9998 gate
    uv = np.stack([k1["x"], k1["y"]], 1).astype(np.float64)
    Xc = np.stack([(uv[:, 0] - K_KITTI[2]) / K_KITTI[0] * depth, (uv[:, 1] - K_KITTI[3]) / K_KITTI[1] * depth, depth], 1)
    R1, t1 = T1[:3, :3].astype(np.float64), T1[:3, 3].astype(np.float64)
    X = (Xc - t1) @ R1
    uv2, z2 = _project(T2, K_KITTI, X)
    k2 = k1.copy()
    k2["x"] = (uv2[:, 0] + rng.normal(0, noise, n)).astype(np.float32)
    k2["y"] = (uv2[:, 1] + rng.normal(0, noise, n)).astype(np.float32)
    k1["x"] += rng.normal(0, noise, n).astype(np.float32)
    k1["y"] += rng.normal(0, noise, n).astype(np.float32)
    vis = (z2 > 0.5) & (k2["x"] > 0) & (k2["x"] < 1241) & (k2["y"] > 0) & (k2["y"] < 376)
    idx1 = np.nonzero(vis)[0].astype(np.int32)
    idx2 = idx1.copy()
    nw = int(len(idx1) * wrong_frac)
    idx2[:nw] = rng.permutation(idx2[:nw])           # wrong pairs
    k2["octave"][idx2[nw: nw + nw // 2]] = 7          # scale-inconsistent pairs
    good = np.ones(len(idx1), bool)
    good[: nw + nw // 2] = False
    good &= depth[idx1] < 1000
    return k1, k2, idx1, idx2, T1, T2, X.astype(np.float32), good


# ------------------------------------------------------------------ CPU: oracle known answers
def test_oracle_svd4_null_vector(oracle):
    rng = np.random.default_rng(7)
    A = rng.standard_normal((200, 4, 4)).astype(np.float32)
    x = rng.standard_normal((200, 4)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    # force a (near) null vector so that the smallest singular value is well separated
    A -= np.einsum("nij,nj,nk->nik", A, x, x).astype(np.float32) * np.float32(0.999)
    vt = oracle.svd4_vt(A)
    ref = np.linalg.svd(A.astype(np.float64))[2]
    for i in range(len(A)):
        np.testing.assert_allclose(vt[i] @ vt[i].T, np.eye(4), atol=5e-6)
        s = np.sign(vt[i, 3] @ ref[i, 3])
        np.testing.assert_allclose(vt[i, 3] * s, ref[i, 3], atol=2e-4)
        # rows sorted by descending singular value
        sv = np.linalg.norm(A[i].astype(np.float64) @ vt[i].T.astype(np.float64), axis=0)
        assert (np.diff(sv) <= 1e-5).all()


def test_oracle_svd4_identity_and_rank_deficient(oracle):
    vt = oracle.svd4_vt(np.eye(4, dtype=np.float32)[None])[0]
    np.testing.assert_array_equal(vt, np.eye(4, dtype=np.float32))       # nothing to rotate, stable order
    A = np.zeros((4, 4), np.float32)
    A[0, 0], A[1, 1], A[2, 2] = 3, 2, 1                                     # null vector = e3
    np.testing.assert_array_equal(np.abs(oracle.svd4_vt(A[None])[0][3]), np.array([0, 0, 0, 1], np.float32))


def test_oracle_triangulation_recovers_points(oracle):
    k1, k2, idx1, idx2, T1, T2, X, good = _two_keyframes(3000, 11)
    x, ok, nok = oracle.triangulate_pairs(k1, k2, idx1, idx2, T1, T2, K_KITTI, K_KITTI)
    assert nok == ok.sum()
    assert ok[good].mean() > 0.9
    assert ok[~good].mean() < 0.12          # mismatches / zero parallax / wrong scale are gated out
    g = good & (ok == 1)
    rel = np.linalg.norm(x[g] - X[idx1[g]], axis=1) / np.linalg.norm(X[idx1[g]], axis=1)
    assert np.median(rel) < 0.02
    assert (x[ok == 0] == 0).all()


def test_oracle_triangulation_gates(oracle):
    """each gate alone: cheirality (point behind), reprojection chi2, scale ratio, parallax"""
    k1, k2, idx1, idx2, T1, T2, X, good = _two_keyframes(400, 13, noise=0.0, wrong_frac=0.0)
    O1 = -T1[:3, :3].T.astype(np.float64) @ T1[:3, 3]
    O2 = -T2[:3, :3].T.astype(np.float64) @ T2[:3, 3]
    r1, r2 = X[idx1] - O1, X[idx1] - O2
    cosp = (r1 * r2).sum(1) / np.linalg.norm(r1, axis=1) / np.linalg.norm(r2, axis=1)
    sel = np.nonzero(good & (cosp < 0.9995))[0][:50]     # well clear of the 0.9998 parallax gate
    i1, i2 = idx1[sel], idx2[sel]
    x, ok, _ = oracle.triangulate_pairs(k1, k2, i1, i2, T1, T2, K_KITTI, K_KITTI)
    assert ok.all()
    # same keyframe twice: zero parallax -> rejected before the SVD
    assert not oracle.triangulate_pairs(k1, k1, i1, i1, T1, T1, K_KITTI, K_KITTI)[1].any()
    # vertical offset of 6 px at level 0 breaks the epipolar geometry -> chi2 gate
    kb = k2.copy()
    kb["y"][i2] += 12
    kb["octave"][:] = 0
    ka = k1.copy()
    ka["octave"][:] = 0
    assert oracle.triangulate_pairs(ka, kb, i1, i2, T1, T2, K_KITTI, K_KITTI)[1].mean() < 0.1
    # octave mismatch (scale 1.2^5 vs distance ratio ~1) -> scale gate
    kc = k2.copy()
    kc["octave"][i2] = 7
    kd = k1.copy()
    kd["octave"][i1] = 0
    assert not oracle.triangulate_pairs(kd, kc, i1, i2, T1, T2, K_KITTI, K_KITTI)[1].any()


# ------------------------------------------------------------------ GPU parity through the C ABI
@pytest.mark.gpu
def test_svd4_parity(hip, oracle):
    rng = np.random.default_rng(21)
    A = rng.standard_normal((4096, 4, 4)).astype(np.float32)
    A[:100] *= np.float32(1e-3)
    A[100:200, :, 3] = 0        # exact null column
    A[200:300, 2] = A[200:300, 1]  # duplicate rows
    v = hip.svd4_null(A)
    e = oracle.svd4_vt(A)[:, 3]
    np.testing.assert_array_equal(v, e)


@pytest.mark.gpu
@pytest.mark.parametrize("n,noise", [(3000, 0.4), (64, 0.0), (1, 0.3)])
def test_triangulate_pairs(hip, oracle, n, noise):
    k1, k2, idx1, idx2, T1, T2, X, good = _two_keyframes(max(n, 40), 31 + n, noise=noise)
    idx1, idx2 = idx1[:n], idx2[:n]
    d = np.zeros((len(k1), 128), np.float32)
    hip.frame_set(4, k1, d, BOUNDS)
    hip.frame_set(5, k2, d, BOUNDS)
    gx, gok, gn = hip.triangulate_pairs(4, 5, idx1, idx2, T1, T2, K_KITTI, K_KITTI)
    ex, eok, en = oracle.triangulate_pairs(k1, k2, idx1, idx2, T1, T2, K_KITTI, K_KITTI)
    np.testing.assert_array_equal(gok, eok)
    np.testing.assert_array_equal(gx, ex)
    assert gn == en
    if n >= 1000:
        assert gn > 0.5 * n


@pytest.mark.gpu
def test_triangulate_pairs_errors(hip):
    k1, d1 = make_frame(10, 3)
    hip.frame_set(4, k1, d1, BOUNDS)
    hip.frame_set(5, k1, d1, BOUNDS)
    T = pose_T()
    with pytest.raises(Exception):
        hip.triangulate_pairs(4, 5, [0, 10], [0, 1], T, T, K_KITTI, K_KITTI)   # index out of range
    x, ok, n = hip.triangulate_pairs(4, 5, np.zeros(0, np.int32), np.zeros(0, np.int32), T, T, K_KITTI, K_KITTI)
    assert n == 0 and len(ok) == 0


# ------------------------------------------------------------------ the batched per-keyframe stage (round 4)
def _kf_with_neighbours(n, n_nb, seed):
    """a current keyframe + n_nb neighbours observing the same 3-D points from poses along a track: keypoints = projections + noise,
    descriptors = perturbed copies, vocabulary nodes from the descriptors' signs (as tests/test_matcher.py's BoW tests)"""
    from tests.test_matcher import _bow_nodes, perturbed_descriptors
    rng = np.random.default_rng(seed)
    Tc = pose_T((0.01, -0.02, 0.005), (0.1, -0.05, 0.3))
    kc, dc = make_frame(n, seed + 1)
    depth = rng.uniform(4, 35, n)
    uv = np.stack([kc["x"], kc["y"]], 1).astype(np.float64)
    Xc = np.stack([(uv[:, 0] - K_KITTI[2]) / K_KITTI[0] * depth, (uv[:, 1] - K_KITTI[3]) / K_KITTI[1] * depth, depth], 1)
    Rc, tc = Tc[:3, :3].astype(np.float64), Tc[:3, 3].astype(np.float64)
    X = (Xc - tc) @ Rc
    out = []
    for b in range(n_nb):
        Tb = pose_T((0.012 + 0.002 * b, -0.03 - 0.004 * b, 0.001 * b), (0.1 - 0.6 - 0.25 * b, -0.02, 0.1 + 0.05 * b))
        uv2, z2 = _project(Tb, K_KITTI, X)
        kb = kc.copy()
        kb["x"] = (uv2[:, 0] + rng.normal(0, 0.4, n)).astype(np.float32)
        kb["y"] = (uv2[:, 1] + rng.normal(0, 0.4, n)).astype(np.float32)
        vis = (z2 > 0.5) & (kb["x"] > 19) & (kb["x"] < 1221) & (kb["y"] > 19) & (kb["y"] < 356)
        kb["x"][~vis] = rng.uniform(19, 1221, (~vis).sum()).astype(np.float32)
        kb["y"][~vis] = rng.uniform(19, 356, (~vis).sum()).astype(np.float32)
        perm = rng.permutation(n)
        kb, db = kb[perm].copy(), perturbed_descriptors(dc[perm], 0.03, seed + 10 + b)
        # F12 of (current, neighbour) as LocalMapping::ComputeF12 builds it (:547-555): K1^-T [t12]x R12 K2^-1
        R1w, t1w, R2w, t2w = Rc, tc, Tb[:3, :3].astype(np.float64), Tb[:3, 3].astype(np.float64)
        R12 = R1w @ R2w.T
        t12 = -R1w @ R2w.T @ t2w + t1w
        tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
        Km = np.array([[K_KITTI[0], 0, K_KITTI[2]], [0, K_KITTI[1], K_KITTI[3]], [0, 0, 1]], np.float64)
        F12 = (np.linalg.inv(Km).T @ tx @ R12 @ np.linalg.inv(Km)).astype(np.float32)
        C2 = R2w @ (-(Rc.T @ tc)) + t2w          # the current camera's centre in the neighbour's frame -> epipole
        ex, ey = np.float32(K_KITTI[0] * C2[0] / C2[2] + K_KITTI[2]), np.float32(K_KITTI[1] * C2[1] / C2[2] + K_KITTI[3])
        out.append(dict(kps=kb, desc=db, T=Tb, F12=F12, ex=ex, ey=ey, nodes=_bow_nodes(db), has=(rng.uniform(size=n) < 0.25).astype(np.uint8)))
    return kc, dc, Tc, _bow_nodes(dc), (rng.uniform(size=n) < 0.4).astype(np.uint8), out


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_nb", [(2000, 6), (300, 20)])
def test_create_map_points_batch_equals_the_per_pair_calls(hip, oracle, n, n_nb):
    """asd_create_map_points_batch (every neighbour in one submission) against asd_match_triangulate + asd_triangulate_pairs per
    neighbour AND against the oracle directly (oracle.match_triangulate = ORBmatcher::SearchForTriangulation :669-822,
    oracle.triangulate_pairs = LocalMapping::CreateNewMapPoints' per-match body :386-519): the same match ids, the same accept
    flags, the same coordinates bit for bit"""
    kc, dc, Tc, nodes_c, has_c, nbs = _kf_with_neighbours(n, n_nb, 900 + n)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set_bow(0, nodes_c)
    for b, d in enumerate(nbs):
        hip.frame_set(1 + b, d["kps"], d["desc"], BOUNDS)
        hip.frame_set_bow(1 + b, d["nodes"])
    m, nm, x, ok = hip.create_map_points_batch(0, n, has_c, Tc, K_KITTI, [dict(slot=1 + b, has_mp=d["has"], F12=d["F12"], ex=d["ex"], ey=d["ey"], Tcw=d["T"], K=K_KITTI)
                                                                      for b, d in enumerate(nbs)])
    total_ok, ofc = 0, None
    for b, d in enumerate(nbs):
        em, en = hip.match_triangulate(0, 1 + b, n, nodes_c, d["nodes"], has_c, d["has"], d["F12"], d["ex"], d["ey"], False)
        np.testing.assert_array_equal(m[b], em)
        assert nm[b] == en
        i1 = np.nonzero(em >= 0)[0].astype(np.int32)
        ex_, eok, _ = hip.triangulate_pairs(0, 1 + b, i1, em[i1], Tc, d["T"], K_KITTI, K_KITTI)
        np.testing.assert_array_equal(ok[b][i1], eok)
        np.testing.assert_array_equal(x[b][i1], ex_)
        assert not ok[b][em < 0].any() and not x[b][em < 0].any()
        total_ok += int(eok.sum())
        # the batch kernels (k_tri_match_batch, k_triangulate_batch) against the oracle itself, not only through the per-pair HIP calls
        if ofc is None:
            ofc = oracle.frame(kc, dc, BOUNDS)
        om, on = oracle.match_triangulate(ofc, oracle.frame(d["kps"], d["desc"], BOUNDS), nodes_c, d["nodes"], has_c, d["has"], d["F12"], d["ex"], d["ey"], False)
        np.testing.assert_array_equal(m[b], om)
        assert nm[b] == on
        o1 = np.nonzero(om >= 0)[0].astype(np.int32)
        ox, ook, _ = oracle.triangulate_pairs(kc, d["kps"], o1, om[o1], Tc, d["T"], K_KITTI, K_KITTI)
        np.testing.assert_array_equal(ok[b][o1], ook)
        np.testing.assert_array_equal(x[b][o1], ox)
    assert nm.sum() > 0.1 * n * n_nb * 0.3 and total_ok > 0


@pytest.mark.gpu
def test_fuse_search_batch_equals_the_per_call_search(hip, oracle, synth):
    """asd_fuse_search_batch over several (keyframe, map point list) pairs against asd_fuse_search per pair and against
    oracle.fuse_search (ORBmatcher::Fuse :825-936, search half) directly"""
    from tests.test_matcher import backproject, perturbed_descriptors
    rng = np.random.default_rng(77)
    K = np.array(synth.KITTI_K, np.float32)
    calls, tabs, exp, oexp = [], [], [], []
    first = 0
    for c in range(7):
        kc, dc = make_frame(2000 if c < 5 else 300, 500 + c)
        T = pose_T((0.01 + 0.003 * c, -0.02, 0.005), (0.1 - 0.2 * c, -0.05, 0.3))
        n_mp = [4000, 1500, 50, 900, 2500, 700, 1][c]
        src = rng.integers(0, len(kc), n_mp)
        uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
        uv[: n_mp // 8] += 2500
        Xw = backproject(T, K, uv, rng.uniform(3, 60, n_mp))
        Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
        nrm = Xw.astype(np.float64) - Ow
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = (nrm + rng.normal(0, 0.4, nrm.shape)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        dist = np.linalg.norm(Xw.astype(np.float64) - Ow, axis=1)
        maxd = (dist * SCALES[kc["octave"][src]] * rng.uniform(0.9, 1.1, n_mp)).astype(np.float32)
        mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
        desc = perturbed_descriptors(dc[src], 0.04, 600 + c)
        valid = (rng.uniform(size=n_mp) < 0.9).astype(np.uint8)
        hip.frame_set(10 + c, kc, dc, BOUNDS)
        exp.append(hip.fuse_search(10 + c, valid, Xw, nrm, mind, maxd, desc, T, K, 3.0))
        oexp.append(oracle.fuse_search(oracle.frame(kc, dc, BOUNDS), valid, Xw, nrm, mind, maxd, desc, T, K, 3.0))
        calls.append(dict(slot_kf=10 + c, first=first, n=n_mp, Tcw=T, K=K))
        tabs.append((valid, Xw, nrm, mind, maxd, desc))
        first += n_mp
    cat = [np.concatenate([t[k] for t in tabs]) for k in range(6)]
    bi, bd = hip.fuse_search_batch(calls, *cat, th=3.0)
    for c, (ei, ed), (oi, od) in zip(calls, exp, oexp):
        np.testing.assert_array_equal(bi[c["first"]: c["first"] + c["n"]], ei)
        np.testing.assert_array_equal(bd[c["first"]: c["first"] + c["n"]], ed)
        np.testing.assert_array_equal(bi[c["first"]: c["first"] + c["n"]], oi)      # k_fuse_batch + the shared search against the oracle itself
        np.testing.assert_array_equal(bd[c["first"]: c["first"] + c["n"]], od)
    assert (bi >= 0).sum() > 2000
    # the map points' descriptors as rows of the descriptor bank instead of 512-byte rows
    hip.bank_put(1000, cat[5])
    ri, rd = hip.fuse_search_batch(calls, *cat[:5], np.arange(1000, 1000 + len(cat[5]), dtype=np.int32), th=3.0)
    np.testing.assert_array_equal(ri, bi)
    np.testing.assert_array_equal(rd, bd)
