"""Frame grid + matchers: oracle self-checks on CPU, HIP vs oracle
bit-exact (indices, match ids, distances) on the GPU."""
import numpy as np
import pytest

BOUNDS = (0.0, 1241.0, 0.0, 376.0)
SCALES = np.float32(1.2) ** np.arange(8)


def make_frame(n, seed):
    """Random keypoints with the extractor's per-level structure + unit descriptors."""
    rng = np.random.default_rng(seed)
    from tests.conftest import load_package
    KP = load_package().capi.KP_DTYPE
    kps = np.zeros(n, KP)
    octv = np.sort(rng.choice(8, n, p=np.array([434, 362, 302, 251, 209, 175, 145, 122]) / 2000.0))
    kps["octave"] = octv
    kps["x"] = rng.uniform(19, 1221, n).astype(np.float32)
    kps["y"] = rng.uniform(19, 356, n).astype(np.float32)
    kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kps["response"] = rng.integers(7, 200, n)
    kps["size"] = 31
    d = rng.standard_normal((n, 128)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return kps, d.astype(np.float32)


def perturbed_descriptors(desc, sigma, seed):
    rng = np.random.default_rng(seed)
    d = desc + rng.standard_normal(desc.shape).astype(np.float32) * np.float32(sigma)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d.astype(np.float32)


def pose_T(rv=(0.01, -0.02, 0.005), t=(0.1, -0.05, 0.3)):
    from tests.conftest import load_package
    s = load_package().synth
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = s._rot(np.array(rv)).astype(np.float32)
    T[:3, 3] = np.array(t, np.float32)
    return T


def backproject(T, K, uv, depth):
    fx, fy, cx, cy = K
    Xc = np.stack([(uv[:, 0] - cx) / fx * depth, (uv[:, 1] - cy) / fy * depth, depth], 1).astype(np.float64)
    R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    return ((Xc - t) @ R).astype(np.float32)  # R^T (Xc - t)


# ------------------------------------------------------------------ oracle self-checks (CPU)
def test_descriptor_distance_order(oracle, synth):
    d = synth.unit_descriptors(6, seed=1)
    for i in range(6):
        for j in range(6):
            acc = np.float32(0)
            for k in range(128):
                diff = np.float32(d[i, k] - d[j, k])
                acc = np.float32(acc + np.float32(diff * diff))
            assert oracle.descriptor_distance(d[i], d[j]) == acc
    M = oracle.dist_matrix(d, d)
    assert (np.diag(M) == 0).all() and (M == M.T).all()


def test_grid_query_vs_bruteforce(oracle):
    kps, desc = make_frame(1500, 5)
    F = oracle.frame(kps, desc, BOUNDS)
    rng = np.random.default_rng(6)
    inv_w, inv_h = np.float32(64) / np.float32(1241), np.float32(48) / np.float32(376)
    cx = np.round((kps["x"] - np.float32(0)) * inv_w).astype(int)
    cy = np.round((kps["y"] - np.float32(0)) * inv_h).astype(int)
    for _ in range(200):
        x, y = np.float32(rng.uniform(-30, 1270)), np.float32(rng.uniform(-30, 400))
        r = np.float32(rng.uniform(1, 60))
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        got = F.features_in_area(x, y, r, lo, hi)
        x0 = max(0, int(np.floor((x - r) * inv_w))); x1 = min(63, int(np.ceil((x + r) * inv_w)))
        y0 = max(0, int(np.floor((y - r) * inv_h))); y1 = min(47, int(np.ceil((y + r) * inv_h)))
        exp = []
        if x0 < 64 and x1 >= 0 and y0 < 48 and y1 >= 0:
            for ix in range(x0, x1 + 1):
                for iy in range(y0, y1 + 1):
                    for i in np.nonzero((cx == ix) & (cy == iy))[0]:
                        o = kps["octave"][i]
                        if lo > 0 or hi >= 0:
                            if o < lo or (hi >= 0 and o > hi):
                                continue
                        if abs(kps["x"][i] - x) < r and abs(kps["y"][i] - y) < r:
                            exp.append(int(i))
        assert got.tolist() == exp


def test_distinctive_descriptor_oracle(oracle, synth):
    base = synth.unit_descriptors(1, seed=9)[0]
    obs = np.stack([perturbed_descriptors(base[None], s, 10 + i)[0] for i, s in enumerate([0.02, 0.3, 0.25, 0.01, 0.4])])
    best = oracle.distinctive_descriptor(obs)
    M = oracle.dist_matrix(obs, obs)
    med = np.sort(M, axis=1)[:, 2]
    assert best == int(np.argmin(med))


def test_oracle_m1_recovers_true_matches(oracle, synth):
    kl, dl = make_frame(1200, 21)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    # current frame = same keypoints shifted by a few px, descriptors slightly perturbed, shuffled
    rng = np.random.default_rng(22)
    perm = rng.permutation(len(kl))
    kc = kl[perm].copy()
    kc["x"] += rng.uniform(-3, 3, len(kc)).astype(np.float32)
    kc["y"] += rng.uniform(-3, 3, len(kc)).astype(np.float32)
    dc = perturbed_descriptors(dl[perm], 0.02, 23)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    uv = np.stack([kc["x"][inv], kc["y"][inv]], 1)
    Xw = backproject(T, K, uv, rng.uniform(5, 40, len(kl)))
    has = (rng.uniform(size=len(kl)) < 0.8).astype(np.uint8)
    cur, last = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
    m, n = oracle.match_project_frame(cur, last, has, Xw, dl, T, K, 15.0, check_ori=False)
    good = sum(1 for j in range(len(kc)) if m[j] >= 0 and m[j] == perm[j])
    assert n == (m >= 0).sum() and good > 0.9 * has.sum() * 0.9


def _contested_m1_case(synth, n, seed):
    """_m1_case-like input in which every map point has a twin projecting onto the same current keypoint, so the
    'already holds a map point' branch (ORBmatcher.cc:1392-1395) decides most matches"""
    kl, dl = make_frame(n, seed)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(seed + 1)
    half = n // 2
    twin = np.concatenate([np.arange(half), np.arange(half)])[:n]      # last keypoint i aims at current keypoint twin[i]
    kc = kl[:half].copy()
    kc["x"] += rng.uniform(-2, 2, half).astype(np.float32)
    kc["y"] += rng.uniform(-2, 2, half).astype(np.float32)
    kc["angle"] = (kc["angle"] + rng.normal(0, 25, half)).astype(np.float32) % np.float32(360)
    dc = perturbed_descriptors(dl[:half], 0.03, seed + 2)
    kl = kl.copy()
    kl["octave"] = kc["octave"][twin]
    kl["angle"] = (kc["angle"][twin] + rng.normal(0, 40, n)).astype(np.float32) % np.float32(360)
    uv = np.stack([kc["x"][twin], kc["y"][twin]], 1)
    Xw = backproject(T, K, uv, rng.uniform(5, 40, n))
    has = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    mp_desc = perturbed_descriptors(dc[twin], 0.03, seed + 3)
    obs = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    return kl, dl, kc, dc, Xw, has, mp_desc, T, K, obs


def _m1_python(oracle, cur, last_kps, cur_kps, has, Xw, mp_desc, dc, T, K, th, check_ori, obs):
    """the reference loop (ORBmatcher.cc:1318-1452) written out in Python over the oracle's grid query and distance"""
    n_cur = len(cur_kps)
    held = [-1] * n_cur
    nm = 0
    hist = [[] for _ in range(30)]
    scale = SCALES.astype(np.float32)
    for i in range(len(last_kps)):
        if not has[i]:
            continue
        x3 = [np.float32(np.float64(np.float32(np.float32(np.float32(T[r, 0] * Xw[i, 0]) + np.float32(T[r, 1] * Xw[i, 1]))
                                               + np.float32(T[r, 2] * Xw[i, 2]))) + np.float64(T[r, 3])) for r in range(3)]
        invz = np.float32(1.0 / np.float64(x3[2]))
        if invz < 0:
            continue
        u = np.float32(np.float32(np.float32(K[0] * x3[0]) * invz) + K[2])
        v = np.float32(np.float32(np.float32(K[1] * x3[1]) * invz) + K[3])
        if u < BOUNDS[0] or u > BOUNDS[1] or v < BOUNDS[2] or v > BOUNDS[3]:
            continue
        o = int(last_kps["octave"][i])
        cand = cur.features_in_area(float(u), float(v), float(np.float32(th) * scale[o]), o - 1, o + 1)
        best, bj = np.float32(100), -1
        for j in cand:
            if held[j] >= 0 and obs[held[j]]:
                continue
            d = oracle.descriptor_distance(mp_desc[i], dc[j])
            if d < best:
                best, bj = d, int(j)
        if best <= np.float32(1.5):   # TH_HIGH
            held[bj] = i
            nm += 1
            if check_ori:
                rot = np.float32(last_kps["angle"][i]) - np.float32(cur_kps["angle"][bj])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360))
                b = int(np.round(np.float32(rot * np.float32(1.0 / 30))))
                if b == 30:
                    b = 0
                hist[b].append(bj)
    if check_ori:
        cnt = [len(h) for h in hist]
        order = sorted(range(30), key=lambda b: -cnt[b])
        m1, m2, m3 = cnt[order[0]], cnt[order[1]], cnt[order[2]]
        keep = {order[0]}
        if m2 >= 0.1 * m1:
            keep.add(order[1])
            if m3 >= 0.1 * m1:
                keep.add(order[2])
        for b in range(30):
            if b not in keep:
                for j in hist[b]:
                    held[j] = -1
                    nm -= 1
    return np.array(held, np.int32), nm


@pytest.mark.parametrize("ori", [False, True])
def test_oracle_m1_overwrites_map_points_without_observations(oracle, synth, ori):
    """ORBmatcher.cc:1392-1395: a keypoint holding a map point with Observations() == 0 is not skipped; the later map
    point overwrites it and both writes count.  The oracle against the loop written out in Python (ties in the top-3
    histogram bins would be order dependent: the seed has none)."""
    n = 160
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, obs = _contested_m1_case(synth, n, 300)
    cur, last = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
    m_all, n_all = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, check_ori=ori)
    m_obs, n_obs = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, check_ori=ori, obs_positive=obs)
    m_one, n_one = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, check_ori=ori,
                                              obs_positive=np.ones(n, np.uint8))
    np.testing.assert_array_equal(m_all, m_one)
    assert n_all == n_one
    assert not np.array_equal(m_obs, m_all)            # the flag vector changes who wins
    if not ori:
        assert n_obs > (m_obs >= 0).sum()              # overwritten writes still count in the return value
        assert n_all == (m_all >= 0).sum()
    pm, pn = _m1_python(oracle, cur, kl, kc, has, Xw, mp_desc, dc, T, K, 15.0, ori, obs)
    np.testing.assert_array_equal(m_obs, pm)
    assert n_obs == pn


# ------------------------------------------------------------------ HIP vs oracle (GPU)
@pytest.mark.gpu
def test_dist_matrix_bit_exact(hip, oracle, synth):
    a, b = synth.unit_descriptors(70, seed=31), synth.unit_descriptors(300, seed=32)
    np.testing.assert_array_equal(hip.dist_matrix(a, b), oracle.dist_matrix(a, b))
    assert hip.dist_matrix(a[:0], b).shape == (0, 300)
    one = hip.dist_matrix(a[:1], b[:1])
    assert one[0, 0] == oracle.descriptor_distance(a[0], b[0])


@pytest.mark.gpu
def test_dist_matrix_full_size(hip, oracle, synth):
    d = synth.unit_descriptors(2000, seed=33)
    M = hip.dist_matrix(d, d)
    assert (np.diag(M) == 0).all() and (M == M.T).all()
    np.testing.assert_array_equal(M, oracle.dist_matrix(d, d))
    ref = ((d[:50, None, :].astype(np.float64) - d[None, :50, :]) ** 2).sum(-1)
    np.testing.assert_allclose(M[:50, :50], ref, atol=1e-5)


@pytest.mark.gpu
def test_grid_queries(hip, oracle):
    kps, desc = make_frame(2000, 41)
    hip.frame_set(0, kps, desc, BOUNDS)
    F = oracle.frame(kps, desc, BOUNDS)
    rng = np.random.default_rng(42)
    for _ in range(300):
        x, y, r = rng.uniform(-30, 1270), rng.uniform(-30, 400), rng.uniform(0.5, 80)
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        np.testing.assert_array_equal(hip.features_in_area(0, x, y, r, lo, hi), F.features_in_area(x, y, r, lo, hi))


def _m1_case(synth, n, seed, shift=3.0, sigma=0.05, frac=0.8):
    kl, dl = make_frame(n, seed)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(seed + 1)
    perm = rng.permutation(n)
    kc = kl[perm].copy()
    kc["x"] += rng.uniform(-shift, shift, n).astype(np.float32)
    kc["y"] += rng.uniform(-shift, shift, n).astype(np.float32)
    kc["angle"] = (kc["angle"] + rng.normal(0, 4, n)).astype(np.float32) % np.float32(360)
    dc = perturbed_descriptors(dl[perm], sigma, seed + 2)
    inv = np.empty_like(perm); inv[perm] = np.arange(n)
    uv = np.stack([kc["x"][inv], kc["y"][inv]], 1)
    Xw = backproject(T, K, uv, rng.uniform(5, 40, n))
    has = (rng.uniform(size=n) < frac).astype(np.uint8)
    mp_desc = perturbed_descriptors(dl, 0.02, seed + 3)
    return kl, dl, kc, dc, Xw, has, mp_desc, T, K


@pytest.mark.gpu
@pytest.mark.parametrize("n,th,ori", [(2000, 15.0, True), (2000, 30.0, True), (500, 15.0, False), (37, 15.0, True)])
def test_match_project_frame(hip, oracle, synth, n, th, ori):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 50 + n)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    got, ng = hip.match_project_frame(0, 1, n, has, Xw, mp_desc, T, K, th, ori)
    exp, ne = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw, mp_desc,
                                         T, K, th, ori)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne
    if n >= 500:
        assert ng > 0.5 * has.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("n,th", [(2000, 120.0), (700, 400.0)])
def test_match_project_frame_long_lists(hip, oracle, synth, n, th):
    """windows of more than 16 grid columns / more than 128 candidates: k_window_search<true> ranks such a list from its copy in
    global memory (slow path), and the replay walks beyond the four heads it keeps in registers"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 150 + n, sigma=0.4)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    obs = (np.random.default_rng(3).uniform(size=n) < 0.7).astype(np.uint8)
    for flags in (None, obs):
        got, ng = hip.match_project_frame(0, 1, n, has, Xw, mp_desc, T, K, th, True, obs_positive=flags)
        exp, ne = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw, mp_desc,
                                             T, K, th, True, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne and ng > 0


def _crowded_case(synth, n, n_cur, seed):
    """n map points aimed at n_cur << n current keypoints packed into a small region with near-identical descriptors: every
    list is long, every head is contested, and most map points end up far down their list or empty-handed"""
    kl, dl = make_frame(n, seed)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(seed + 1)
    kc, _ = make_frame(n_cur, seed + 5)
    kc["x"] = rng.uniform(600, 660, n_cur).astype(np.float32)
    kc["y"] = rng.uniform(170, 210, n_cur).astype(np.float32)
    kc["octave"] = np.sort(rng.integers(2, 5, n_cur))
    base = dl[0]
    dc = perturbed_descriptors(np.repeat(base[None], n_cur, 0), 0.02, seed + 2)
    aim = rng.integers(0, n_cur, n)
    kl = kl.copy()
    kl["octave"] = kc["octave"][aim]
    uv = np.stack([kc["x"][aim], kc["y"][aim]], 1) + rng.uniform(-3, 3, (n, 2)).astype(np.float32)
    Xw = backproject(T, K, uv, rng.uniform(5, 40, n))
    has = np.ones(n, np.uint8)
    mp_desc = perturbed_descriptors(np.repeat(base[None], n, 0), 0.02, seed + 3)
    return kl, dl, kc, dc, Xw, has, mp_desc, T, K, aim


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_cur,th", [(2000, 60, 15.0), (3000, 300, 10.0), (500, 7, 30.0)])
def test_match_project_frame_crowded(hip, oracle, synth, n, n_cur, th):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, _ = _crowded_case(synth, n, n_cur, 400 + n)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    obs = (np.random.default_rng(5).uniform(size=n) < 0.6).astype(np.uint8)
    for flags, ori in ((None, False), (obs, True), (np.ones(n, np.uint8), True)):
        got, ng = hip.match_project_frame(0, 1, n_cur, has, Xw, mp_desc, T, K, th, ori, obs_positive=flags)
        exp, ne = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw, mp_desc,
                                             T, K, th, ori, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne


@pytest.mark.gpu
@pytest.mark.parametrize("n_mp,n_cur,th", [(3000, 200, 5.0), (2500, 40, 12.0)])
def test_match_project_points_crowded(hip, oracle, synth, n_mp, n_cur, th):
    """the local-map search with every keypoint contested by many map points: second-best / ratio decisions deep in the lists"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, aim = _crowded_case(synth, n_mp, n_cur, 700 + n_mp)
    hip.frame_set(0, kc, dc, BOUNDS)
    F = oracle.frame(kc, dc, BOUNDS)
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    nrm = Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(nrm, axis=1)
    nrm = (nrm / dist[:, None]).astype(np.float32)
    maxd = (dist * SCALES[kc["octave"][aim]]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    in_view, proj, level, vc = hip.frustum(0, Xw, nrm, mind, maxd, T, K)
    assert in_view.sum() > 0.8 * n_mp
    rng = np.random.default_rng(9)
    occupied = (rng.uniform(size=n_cur) < 0.2).astype(np.uint8)
    obs = (rng.uniform(size=n_mp) < 0.6).astype(np.uint8)
    for flags in (None, obs):
        got, ng = hip.match_project_points(0, n_cur, in_view, proj, level, vc, mp_desc, occupied, th, 0.8, obs_positive=flags)
        exp, ne = oracle.match_project_points(F, in_view, proj, level, vc, mp_desc, occupied, th, 0.8, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne
        assert not ((got >= 0) & (occupied > 0)).any()


def _few_long_lists_case(synth, seed, n_bg=500, n_q=48, n_cl=160):
    """an ordinary small frame (short lists) plus ONE crowd: n_cl current keypoints with near-identical descriptors inside a single grid
    column, contested by n_q map points -- those few lists take k_window_search<true>'s slow path (more than 64 items in a column),
    whose sorted copies live in the second half of the candidate buffers, while the frame's total stays far below the replay's LDS
    staging capacity (the replay then read such lists from the LDS copy, where they are not)"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n_bg, seed, frac=1.0)
    rng = np.random.default_rng(seed + 9)
    kcl, _ = make_frame(n_cl, seed + 5)
    kcl["x"] = rng.uniform(592.5, 610.0, n_cl).astype(np.float32)      # round(x * 64 / 1241) == 31 for all of them
    kcl["y"] = rng.uniform(170, 215, n_cl).astype(np.float32)
    kcl["octave"] = rng.integers(2, 5, n_cl)
    base = dl[0]
    dcl = perturbed_descriptors(np.repeat(base[None], n_cl, 0), 0.02, seed + 6)
    kq, dq = make_frame(n_q, seed + 7)
    aim = rng.integers(0, n_cl, n_q)
    kq["octave"] = kcl["octave"][aim]
    uvq = (np.stack([kcl["x"][aim], kcl["y"][aim]], 1) + rng.uniform(-3, 3, (n_q, 2))).astype(np.float32)
    Xq = backproject(T, K, uvq, rng.uniform(5, 40, n_q))
    mpq = perturbed_descriptors(np.repeat(base[None], n_q, 0), 0.02, seed + 8)
    # current frame: level-major like the extractor's output; last frame: the crowd's map points scattered among the others
    oc = np.argsort(np.concatenate([kc["octave"], kcl["octave"]]), kind="stable")
    kc_all, dc_all = np.concatenate([kc, kcl])[oc], np.concatenate([dc, dcl])[oc]
    ol = rng.permutation(n_bg + n_q)
    kl_all, dl_all = np.concatenate([kl, kq])[ol], np.concatenate([dl, dq])[ol]
    Xw_all, mp_all = np.concatenate([Xw, Xq])[ol], np.concatenate([mp_desc, mpq])[ol]
    return kl_all, dl_all, kc_all, dc_all, Xw_all, np.ones(n_bg + n_q, np.uint8), mp_all, T, K


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1300, 1301])
def test_match_project_frame_few_long_lists_in_a_small_frame(hip, oracle, synth, seed):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _few_long_lists_case(synth, seed)
    n_cur = len(kc)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    oc, ol = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
    obs = (np.random.default_rng(5).uniform(size=len(kl)) < 0.7).astype(np.uint8)
    for flags, ori in ((None, False), (None, False), (obs, True), (obs, True)):   # twice each: the second call's staging capacity follows the first call's total
        got, ng = hip.match_project_frame(0, 1, n_cur, has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
        exp, ne = oracle.match_project_frame(oc, ol, has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne and ng > 300


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1400, 1401])
def test_match_project_points_few_long_lists_in_a_small_frame(hip, oracle, synth, seed):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _few_long_lists_case(synth, seed)
    n_cur = len(kc)
    hip.frame_set(0, kc, dc, BOUNDS)
    F = oracle.frame(kc, dc, BOUNDS)
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    nrm = Xw.astype(np.float64) - Ow
    dist = np.linalg.norm(nrm, axis=1)
    nrm = (nrm / dist[:, None]).astype(np.float32)
    maxd = (dist * SCALES[kl["octave"]]).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    in_view, proj, level, vc = hip.frustum(0, Xw, nrm, mind, maxd, T, K)
    assert in_view.sum() > 0.8 * len(kl)
    rng = np.random.default_rng(9)
    occupied = (rng.uniform(size=n_cur) < 0.1).astype(np.uint8)
    obs = (rng.uniform(size=len(kl)) < 0.6).astype(np.uint8)
    for flags in (None, None, obs, obs):
        got, ng = hip.match_project_points(0, n_cur, in_view, proj, level, vc, mp_desc, occupied, 3.0, 0.8, obs_positive=flags)
        exp, ne = oracle.match_project_points(F, in_view, proj, level, vc, mp_desc, occupied, 3.0, 0.8, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne and ng > 0
        assert not ((got >= 0) & (occupied > 0)).any()


@pytest.mark.gpu
def test_match_project_frame_degenerate(hip, oracle, synth):
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, 300, 77)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    got, ng = hip.match_project_frame(0, 1, 300, np.zeros(300, np.uint8), Xw, mp_desc, T, K, 15.0, True)
    assert ng == 0 and (got == -1).all()
    # points behind the camera / outside the image are skipped
    Xw2 = Xw.copy(); Xw2[:100, 2] -= 500
    got, ng = hip.match_project_frame(0, 1, 300, has, Xw2, mp_desc, T, K, 15.0, True)
    exp, ne = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw2, mp_desc,
                                         T, K, 15.0, True)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne


@pytest.mark.gpu
@pytest.mark.parametrize("n_mp,th", [(6000, 1.0), (3000, 5.0), (10, 1.0)])
def test_frustum_and_match_project_points(hip, oracle, synth, n_mp, th):
    kc, dc = make_frame(2000, 90)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(91 + n_mp)
    src = rng.integers(0, 2000, n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-2, 2, (n_mp, 2)).astype(np.float32)
    uv[: n_mp // 10] += 3000  # some outside the image
    depth = rng.uniform(3, 60, n_mp)
    Xw = backproject(T, K, uv, depth)
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    normal = Xw.astype(np.float64) - Ow
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    normal = (normal + rng.normal(0, 0.3, normal.shape)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    dist = np.linalg.norm(Xw.astype(np.float64) - Ow, axis=1)
    lvl = kc["octave"][src]
    maxd = (dist * SCALES[lvl] * rng.uniform(0.9, 1.1, n_mp)).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    hip.frame_set(0, kc, dc, BOUNDS)
    F = oracle.frame(kc, dc, BOUNDS)
    g = hip.frustum(0, Xw, normal, mind, maxd, T, K)
    e = oracle.frustum(F, Xw, normal, mind, maxd, T, K)
    for a, b in zip(g, e):
        np.testing.assert_array_equal(a, b)
    in_view, proj, level, vc = g
    assert 0.3 * n_mp < in_view.sum() <= n_mp
    desc = perturbed_descriptors(dc[src], 0.05, 92)
    occupied = (rng.uniform(size=2000) < 0.1).astype(np.uint8)
    got, ng = hip.match_project_points(0, 2000, in_view, proj, level, vc, desc, occupied, th, 0.8)
    exp, ne = oracle.match_project_points(F, in_view, proj, level, vc, desc, occupied, th, 0.8)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne
    assert not ((got >= 0) & (occupied > 0)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("n,ori", [(2000, True), (2000, False), (160, True)])
def test_match_project_frame_obs_positive(hip, oracle, synth, n, ori):
    """mixed Observations() flags (ORBmatcher.cc:1392-1395): overwrites, double counting and double histogram entries"""
    kl, dl, kc, dc, Xw, has, mp_desc, T, K, obs = _contested_m1_case(synth, n, 300 + n)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    cur, last = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
    for flags in (obs, np.zeros(n, np.uint8), None):
        got, ng = hip.match_project_frame(0, 1, len(kc), has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
        exp, ne = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne
    hip.bank_put(0, mp_desc)
    got, ng = hip.match_project_frame_bank(0, 1, len(kc), has, Xw, np.arange(n, dtype=np.int32), T, K, 15.0, ori, obs_positive=obs)
    exp, ne = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=obs)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne


@pytest.mark.gpu
def test_host_and_device_replay_agree(pkg, oracle, synth, monkeypatch):
    """The claim / ratio / histogram replay runs on the device (k_resolve: the serial recurrence solved by fixed-point
    iteration) by default and on the host over the copied-back candidate lists with ASD_MATCH_REPLAY=host (also the
    fallback for very large inputs): both against the oracle on contested inputs, all-positive and mixed flags."""
    monkeypatch.setenv("ASD_MATCH_REPLAY", "host")
    H = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376)
    monkeypatch.delenv("ASD_MATCH_REPLAY")
    D = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376)
    try:
        for n, ori in ((2000, True), (700, False)):
            kl, dl, kc, dc, Xw, has, mp_desc, T, K, obs = _contested_m1_case(synth, n, 900 + n)
            cur, last = oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS)
            for flags in (None, obs):
                exp, ne = oracle.match_project_frame(cur, last, has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
                for dev in (H, D):
                    dev.frame_set(0, kc, dc, BOUNDS)
                    dev.frame_set(1, kl, dl, BOUNDS)
                    got, ng = dev.match_project_frame(0, 1, len(kc), has, Xw, mp_desc, T, K, 15.0, ori, obs_positive=flags)
                    np.testing.assert_array_equal(got, exp)
                    assert ng == ne
        # wide windows: lists of 100+ candidates, several per thread beyond the register-resident part
        kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, 2000, 950)
        exp, ne = oracle.match_project_frame(oracle.frame(kc, dc, BOUNDS), oracle.frame(kl, dl, BOUNDS), has, Xw, mp_desc, T, K, 60.0, True)
        for dev in (H, D):
            dev.frame_set(0, kc, dc, BOUNDS)
            dev.frame_set(1, kl, dl, BOUNDS)
            got, ng = dev.match_project_frame(0, 1, 2000, has, Xw, mp_desc, T, K, 60.0, True)
            np.testing.assert_array_equal(got, exp)
            assert ng == ne
    finally:
        H.close()
        D.close()


@pytest.mark.gpu
def test_match_project_points_obs_positive(hip, oracle, synth):
    """M2 with map points that have no observations (ORBmatcher.cc:86-88): the keypoint stays available"""
    kc, dc = make_frame(1500, 95)
    rng = np.random.default_rng(96)
    n_mp = 4000
    src = rng.integers(0, 1500, n_mp)                     # ~2.7 map points per keypoint
    proj = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
    level = kc["octave"][src].astype(np.int32)
    vc = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    in_view = (rng.uniform(size=n_mp) < 0.9).astype(np.uint8)
    desc = perturbed_descriptors(dc[src], 0.04, 97)
    occupied = (rng.uniform(size=1500) < 0.1).astype(np.uint8)
    obs = (rng.uniform(size=n_mp) < 0.5).astype(np.uint8)
    hip.frame_set(0, kc, dc, BOUNDS)
    F = oracle.frame(kc, dc, BOUNDS)
    res = []
    for flags in (obs, None):
        got, ng = hip.match_project_points(0, 1500, in_view, proj, level, vc, desc, occupied, 1.0, 0.8, obs_positive=flags)
        exp, ne = oracle.match_project_points(F, in_view, proj, level, vc, desc, occupied, 1.0, 0.8, obs_positive=flags)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne
        res.append((got, ng))
    assert not np.array_equal(res[0][0], res[1][0])
    assert res[0][1] > 2 * (res[0][0] >= 0).sum()         # overwritten writes are still counted (twice each)
    assert res[1][1] == 2 * (res[1][0] >= 0).sum()


@pytest.mark.gpu
def test_match_init(hip, oracle, synth):
    k1, d1 = make_frame(4000, 101)
    rng = np.random.default_rng(102)
    perm = rng.permutation(4000)
    k2 = k1[perm].copy()
    k2["x"] += rng.uniform(-20, 20, 4000).astype(np.float32)
    k2["y"] += rng.uniform(-8, 8, 4000).astype(np.float32)
    d2 = perturbed_descriptors(d1[perm], 0.03, 103)
    hip.frame_set(2, k1, d1, BOUNDS)
    hip.frame_set(3, k2, d2, BOUNDS)
    prev = np.stack([k1["x"], k1["y"]], 1)
    got, ng, pm_g = hip.match_init(2, 3, prev, 100, 0.9, True)
    exp, ne, pm_e = oracle.match_init(oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS), prev, 100, 0.9, True)
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(pm_g, pm_e)
    assert ng == ne and ng > 200
    assert (got[k1["octave"] > 0] == -1).all()  # level-0 keypoints only (ORBmatcher.cc:434-436)


@pytest.mark.gpu
def test_distinctive_descriptor(hip, oracle, synth):
    base = synth.unit_descriptors(1, seed=111)[0]
    for n in (1, 2, 7, 30):
        obs = np.stack([perturbed_descriptors(base[None], 0.05 * (1 + (i * 7) % 5), 112 + i)[0] for i in range(n)])
        assert hip.distinctive_descriptor(obs) == oracle.distinctive_descriptor(obs)


@pytest.mark.gpu
def test_matchers_on_real_extraction(hip, oracle, synth):
    """End to end on the synthetic stream: extract two frames, adopt device-resident descriptors
    (desc=None), track frame 1 against frame 0 with a planar scene model."""
    K = np.array(synth.KITTI_K, np.float32)
    k0, d0 = hip.extract(synth.scene_frame(0))
    hip.frame_set(1, k0, None, BOUNDS)
    k1, d1 = hip.extract(synth.scene_frame(1))
    hip.frame_set(0, k1, None, BOUNDS)
    T = np.eye(4, dtype=np.float32)
    # frame t+1 = frame t shifted by (-3, -0.2) px and zoomed by ~0.3 % about the image centre
    z = 1.003 / 1.0
    uv = np.stack([(k0["x"] - 620.5) * z + 620.5 - 3 * z, (k0["y"] - 188) * z + 188 - 0.2 * z], 1).astype(np.float32)
    Xw = backproject(T, K, uv, np.full(len(k0), 20.0))
    has = np.ones(len(k0), np.uint8)
    got, ng = hip.match_project_frame(0, 1, len(k1), has, Xw, d0, T, K, 15.0, True)
    exp, ne = oracle.match_project_frame(oracle.frame(k1, d1, BOUNDS), oracle.frame(k0, d0, BOUNDS), has, Xw, d0, T, K,
                                         15.0, True)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne and ng > 300


@pytest.mark.gpu
def test_cpp_host_mirror_example(synth, tmp_path):
    """asd-slam_amd/host: the C++ mirror of ORBextractor / ORBmatcher / Optimizer drives the C ABI end to end."""
    import os, subprocess
    from tests.conftest import ROOT
    exe = os.path.join(ROOT, "asd-slam_amd", "host", "example_track")
    assert os.path.exists(exe), "host example not built (run __graft_entry__.build())"
    with open(tmp_path / "w.bin", "wb") as f:
        for w, m, v in synth.asdnet_weights(0):
            f.write(w.tobytes()); f.write(m.tobytes()); f.write(v.tobytes())
    for t in (0, 1):
        synth.scene_frame(t).tofile(tmp_path / f"f{t}.raw")
    p = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "f0.raw"), str(tmp_path / "f1.raw"), "1241", "376"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "matches=" in p.stdout


# ------------------------------------------------------------------ BoW-guided matchers (M3, M4 triangulation)
def _bow_nodes(desc, nbits=6):
    """stand-in for the DBoW2 transform (vocabulary file absent): node = sign pattern of the first dims, so that
    true matches mostly share a node"""
    bits = (desc[:, :nbits] > 0).astype(np.int64)
    return (bits * (1 << np.arange(nbits))).sum(1).astype(np.int32)


def _two_views(n, seed):
    k1, d1 = make_frame(n, seed)
    rng = np.random.default_rng(seed + 1)
    perm = rng.permutation(n)
    k2 = k1[perm].copy()
    # pure horizontal translation between the cameras -> epipolar lines are image rows: F = [t]_x with t = (1,0,0)
    k2["x"] = np.clip(k2["x"] + rng.uniform(5, 60, n).astype(np.float32), 19, 1221)
    k2["y"] += rng.normal(0, 0.3, n).astype(np.float32)
    k2["angle"] = (k2["angle"] + rng.normal(0, 3, n)).astype(np.float32) % np.float32(360)
    d2 = perturbed_descriptors(d1[perm], 0.03, seed + 2)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    return k1, d1, k2, d2, perm, F12


def test_oracle_bow_and_triangulation_recover_matches(oracle):
    k1, d1, k2, d2, perm, F12 = _two_views(1500, 201)
    n1, n2 = _bow_nodes(d1), _bow_nodes(d2)
    has = np.ones(1500, np.uint8)
    f1, f2 = oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS)
    m, n = oracle.match_bow(f1, f2, n1, n2, has, 0.7, True)
    inv = np.empty_like(perm); inv[perm] = np.arange(1500)
    good = sum(1 for j in range(1500) if m[j] >= 0 and m[j] == perm[j])
    assert n == (m >= 0).sum() and good > 0.4 * 1500 and good >= 0.98 * n
    none = np.zeros(1500, np.uint8)
    m12, nt = oracle.match_triangulate(f1, f2, n1, n2, none, none, F12, -5000.0, 188.0, False)
    good = sum(1 for i in range(1500) if m12[i] >= 0 and perm[m12[i]] == i)
    assert nt == (m12 >= 0).sum() and good > 0.4 * 1500 and good >= 0.95 * nt


@pytest.mark.gpu
@pytest.mark.parametrize("n,ori", [(2000, True), (300, False)])
def test_match_bow(hip, oracle, n, ori):
    k1, d1, k2, d2, perm, _ = _two_views(n, 210 + n)
    n1, n2 = _bow_nodes(d1), _bow_nodes(d2)
    n1[::17] = -1  # keypoints the vocabulary did not place
    rng = np.random.default_rng(5)
    has = (rng.uniform(size=n) < 0.7).astype(np.uint8)
    hip.frame_set(4, k1, d1, BOUNDS)
    hip.frame_set(5, k2, d2, BOUNDS)
    got, ng = hip.match_bow(4, 5, n, n1, n2, has, 0.7, ori)
    exp, ne = oracle.match_bow(oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS), n1, n2, has, 0.7, ori)
    np.testing.assert_array_equal(got, exp)
    assert ng == ne and ng > 0.3 * has.sum()
    # disjoint node sets -> nothing to match
    got, ng = hip.match_bow(4, 5, n, n1 * 0 + 1, n2 * 0 + 2, has, 0.7, ori)
    assert ng == 0 and (got == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,ori", [(2000, False), (500, True)])
def test_match_triangulate(hip, oracle, n, ori):
    k1, d1, k2, d2, perm, F12 = _two_views(n, 230 + n)
    n1, n2 = _bow_nodes(d1), _bow_nodes(d2)
    rng = np.random.default_rng(6)
    has1 = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    has2 = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    hip.frame_set(4, k1, d1, BOUNDS)
    hip.frame_set(5, k2, d2, BOUNDS)
    for ex, ey in ((-5000.0, 188.0), (600.0, 188.0)):  # far epipole; epipole inside the image (rejects close points)
        got, ng = hip.match_triangulate(4, 5, n, n1, n2, has1, has2, F12, ex, ey, ori)
        exp, ne = oracle.match_triangulate(oracle.frame(k1, d1, BOUNDS), oracle.frame(k2, d2, BOUNDS), n1, n2, has1, has2,
                                           F12, ex, ey, ori)
        np.testing.assert_array_equal(got, exp)
        assert ng == ne
    assert (got[has1 > 0] == -1).all()


@pytest.mark.gpu
def test_descriptor_bank_variants(hip, oracle, synth):
    """row ids into the device-resident descriptor bank give the same matches as passing the descriptors"""
    n = 1500
    kl, dl, kc, dc, Xw, has, mp_desc, T, K = _m1_case(synth, n, 311)
    hip.frame_set(0, kc, dc, BOUNDS)
    hip.frame_set(1, kl, dl, BOUNDS)
    rng = np.random.default_rng(3)
    rows = rng.permutation(5000)[:n].astype(np.int32)      # scattered bank rows
    bank = np.zeros((5000, 128), np.float32)
    bank[rows] = mp_desc
    hip.bank_put(0, bank)
    a, na = hip.match_project_frame(0, 1, n, has, Xw, mp_desc, T, K, 15.0, True)
    b, nb = hip.match_project_frame_bank(0, 1, n, has, Xw, rows, T, K, 15.0, True)
    np.testing.assert_array_equal(a, b)
    assert na == nb
    # M2 through the bank, with the bank filled device-to-device from a frame
    hip.bank_put_from_frame(1, 6000, n)
    in_view = np.ones(n, np.uint8)
    proj = np.stack([kc["x"], kc["y"]], 1)[rng.permutation(n)]
    level = rng.integers(0, 8, n).astype(np.int32)
    vc = rng.uniform(0.9, 1.0, n).astype(np.float32)
    occ = np.zeros(n, np.uint8)
    a, na = hip.match_project_points(0, n, in_view, proj, level, vc, dl, occ, 1.0, 0.8)
    b, nb = hip.match_project_points_bank(0, n, in_view, proj, level, vc, np.arange(6000, 6000 + n, dtype=np.int32), occ, 1.0, 0.8)
    np.testing.assert_array_equal(a, b)
    assert na == nb
    with pytest.raises(Exception):
        hip.match_project_frame_bank(0, 1, n, has, Xw, rows + 10_000_000, T, K, 15.0, True)


@pytest.mark.gpu
@pytest.mark.parametrize("n_mp,th", [(4000, 3.0), (50, 3.0)])
def test_fuse_search(hip, oracle, synth, n_mp, th):
    """ORBmatcher::Fuse search half: best keypoint per candidate map point, bit-exact ids and distances"""
    kc, dc = make_frame(2000, 401)
    K = np.array(synth.KITTI_K, np.float32)
    T = pose_T()
    rng = np.random.default_rng(402 + n_mp)
    src = rng.integers(0, 2000, n_mp)
    uv = np.stack([kc["x"][src], kc["y"][src]], 1) + rng.uniform(-1.5, 1.5, (n_mp, 2)).astype(np.float32)
    uv[: n_mp // 8] += 2500
    Xw = backproject(T, K, uv, rng.uniform(3, 60, n_mp))
    Ow = -(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))
    normal = Xw.astype(np.float64) - Ow
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    normal = (normal + rng.normal(0, 0.4, normal.shape)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    dist = np.linalg.norm(Xw.astype(np.float64) - Ow, axis=1)
    maxd = (dist * SCALES[kc["octave"][src]] * rng.uniform(0.9, 1.1, n_mp)).astype(np.float32)
    mind = (maxd / np.float32(SCALES[7])).astype(np.float32)
    desc = perturbed_descriptors(dc[src], 0.04, 403)
    valid = (rng.uniform(size=n_mp) < 0.9).astype(np.uint8)
    hip.frame_set(6, kc, dc, BOUNDS)
    gi, gd = hip.fuse_search(6, valid, Xw, normal, mind, maxd, desc, T, K, th)
    ei, ed = oracle.fuse_search(oracle.frame(kc, dc, BOUNDS), valid, Xw, normal, mind, maxd, desc, T, K, th)
    np.testing.assert_array_equal(gi, ei)
    np.testing.assert_array_equal(gd, ed)
    assert (gi[valid == 0] == -1).all()
    if n_mp > 1000:
        assert (gi >= 0).sum() > 0.3 * n_mp


@pytest.mark.gpu
def test_distinctive_descriptor_batch(hip, oracle, synth):
    """many map points in one call: same pick as MapPoint::ComputeDistinctiveDescriptors per point, including sets
    larger than one workgroup stages (> 64 observations), singletons, pairs and exact duplicates (median ties)"""
    rng = np.random.default_rng(77)
    sizes = [1, 2, 3, 2, 30, 64, 65, 7, 100, 5, 64, 1] + list(rng.integers(1, 40, 300))
    base = synth.unit_descriptors(len(sizes), seed=78)
    sets = []
    for k, n in enumerate(sizes):
        d = perturbed_descriptors(np.repeat(base[k:k + 1], n, 0), 0.08, 79 + k)
        if n >= 4 and k % 5 == 0:
            d[1] = d[0]                       # identical observations -> equal rows / medians
            d[3] = d[2]
        sets.append(d)
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    got = hip.distinctive_descriptor_batch(start, np.concatenate(sets))
    exp = np.array([oracle.distinctive_descriptor(d) for d in sets], np.int32)
    np.testing.assert_array_equal(got, exp)
    with pytest.raises(Exception):
        hip.distinctive_descriptor_batch(np.array([0, 3, 3], np.int32), np.concatenate(sets)[:3])   # empty set
