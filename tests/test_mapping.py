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
