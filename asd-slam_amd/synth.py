"""Seeded synthetic inputs for the hot path.

Trained ASDNet weights, KITTI images and the BoW vocabulary are absent from the
reference tree (.MISSING_LARGE_BLOBS:1-4), so every test / bench input is generated
here from numpy's PCG64 streams, which are bit-stable across platforms.  Nothing in
this file computes results of the hot path; it only makes inputs.
"""
import numpy as np

# (cout, cin, k, stride, pad, relu) -- reference ASDNet/ASDNet/ASDNet.py:334-356
ASDNET_LAYERS = (
    (32, 1, 3, 1, 1, True),
    (32, 32, 3, 1, 1, True),
    (64, 32, 3, 2, 1, True),
    (64, 64, 3, 1, 1, True),
    (128, 64, 3, 2, 1, True),
    (128, 128, 3, 1, 1, True),
    (128, 128, 8, 1, 0, False),
)
ASDNET_BN_EPS = 1e-5  # nn.BatchNorm2d default, ASDNet.py:336
ASDNET_MACS_PER_PATCH = 39_092_224
ASDNET_FLOP_PER_PATCH = 2 * ASDNET_MACS_PER_PATCH

KITTI_W, KITTI_H = 1241, 376
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)  # cameraconfig/KITTI/kitti00-02.txt:1


def asdnet_weights(seed=0):
    """Seeded stand-in for the missing bestmodel_c.pt.

    Returns a list of 7 (w[cout,cin,k,k] f32, bn_mean[cout] f32, bn_var[cout] f32).
    He-scaled normal conv weights; BN running stats are non-trivial so that a wrong
    BN fold cannot hide behind mean=0 / var=1.
    """
    rng = np.random.default_rng(seed)
    out = []
    for cout, cin, k, _s, _p, _r in ASDNET_LAYERS:
        fan_in = cin * k * k
        w = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        mean = (rng.standard_normal(cout) * 0.2).astype(np.float32)
        var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        out.append((w, mean, var))
    return out


def random_patches(n, seed=1):
    """n x 32 x 32 u8 patches: smooth blobs + noise; patch 0 is constant (std = 0)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (n, 8, 8)).astype(np.float32)
    up = np.kron(base, np.ones((4, 4), np.float32))
    up += rng.standard_normal((n, 32, 32)).astype(np.float32) * 12.0
    p = np.clip(np.rint(up), 0, 255).astype(np.uint8)
    if n > 0:
        p[0, :, :] = 77
    return p


def scene_frame(t, seed=20, w=KITTI_W, h=KITTI_H, _cache={}):
    """Frame t of the SURVEY §8(d) synthetic stream: a textured scene rendered once at
    2x size, cropped with a (3t, 0.2t) px drift and a 0.3 %/frame zoom, u8 h x w."""
    key = (seed, w, h)
    if key not in _cache:
        rng = np.random.default_rng(seed)
        W2, H2 = 2 * w + 800, 2 * h + 200
        img = np.full((H2, W2), 128.0, np.float32)
        nrect = 6000
        xs = rng.integers(0, W2, nrect)
        ys = rng.integers(0, H2, nrect)
        ws = rng.integers(4, 41, nrect)
        hs = rng.integers(4, 41, nrect)
        gs = rng.integers(20, 236, nrect)
        for x, y, ww, hh, g in zip(xs, ys, ws, hs, gs):
            img[y:y + hh, x:x + ww] = g
        img += rng.standard_normal(img.shape).astype(np.float32) * 6.0
        _cache[key] = np.clip(img, 0, 255)
    big = _cache[key]
    H2, W2 = big.shape
    zoom = 1.0 + 0.003 * t
    # sample grid (nearest-neighbour on the 2x render, then 2x2 box) -- cheap and deterministic
    sx = (np.arange(w, dtype=np.float64) - w / 2) * (2.0 / zoom) + w + 3.0 * t * 2
    sy = (np.arange(h, dtype=np.float64) - h / 2) * (2.0 / zoom) + h + 0.2 * t * 2
    x0 = np.clip(np.floor(sx).astype(np.int64), 0, W2 - 2)
    y0 = np.clip(np.floor(sy).astype(np.int64), 0, H2 - 2)
    a = big[np.ix_(y0, x0)] + big[np.ix_(y0, x0 + 1)] + big[np.ix_(y0 + 1, x0)] + big[np.ix_(y0 + 1, x0 + 1)]
    return np.clip(np.rint(a * 0.25), 0, 255).astype(np.uint8)


def unit_descriptors(n, seed=3, dim=128):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, dim)).astype(np.float32)
    d /= np.sqrt((d.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    return d.astype(np.float32)


def _rot(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rot_to_quat(R):
    """Rotation matrix -> unit quaternion (x, y, z, w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def ba_problem(n_free=24, n_fixed=12, n_points=6000, obs_per_point=5, seed=1,
               pix_noise=0.3, outlier_frac=0.02, pose_sigma=0.01, point_sigma=0.05):
    """SURVEY §8(d) nominal LocalBA problem.  Poses on a forward track (0.8 m apart);
    fixed poses come first in id order (older keyframes), like the reference's id-sorted
    vertex set (sparse_optimizer.cpp:166-190).

    Returns dict of flat arrays in the C-ABI layout of include/asd_slam.h:
      poses  [P,7] f64  (qx,qy,qz,qw,tx,ty,tz) world->camera, fixed [P] u8,
      points [L,3] f64, edges: e_point [E] i32, e_pose [E] i32, e_obs [E,2] f64,
      e_info [E] f64 (invSigma2), K (fx,fy,cx,cy).
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = KITTI_K
    P = n_free + n_fixed
    Rs, ts = [], []
    for i in range(P):
        R = _rot(rng.standard_normal(3) * 0.01)
        c = np.array([rng.standard_normal() * 0.05, rng.standard_normal() * 0.02, 0.8 * i])
        Rs.append(R)
        ts.append(-R @ c)
    pts, e_point, e_pose, e_obs, e_info = [], [], [], [], []
    sig2 = 1.2 ** (2 * np.arange(8))
    l = 0
    attempts = 0
    while l < n_points and attempts < 50 * n_points:
        attempts += 1
        first = int(rng.integers(0, P - obs_per_point + 1))
        zc = 0.8 * first + rng.uniform(8, 28)
        X = np.array([rng.uniform(-10, 10), rng.uniform(-3, 3), zc])
        obs = []
        for p in range(first, first + obs_per_point):
            Xc = Rs[p] @ X + ts[p]
            if Xc[2] < 1.0:
                break
            u = fx * Xc[0] / Xc[2] + cx
            v = fy * Xc[1] / Xc[2] + cy
            if not (0 <= u < KITTI_W and 0 <= v < KITTI_H):
                break
            obs.append((p, u, v))
        if len(obs) < 2:
            continue
        for p, u, v in obs:
            du, dv = rng.uniform(-pix_noise, pix_noise, 2)
            if rng.uniform() < outlier_frac:
                du, dv = rng.choice([-20.0, 20.0], 2)
            lvl = int(rng.integers(0, 8))
            e_point.append(l)
            e_pose.append(p)
            # the reference stores keypoints as float32 (cv::KeyPoint), Optimizer.cc:546-547
            e_obs.append((np.float32(u + du), np.float32(v + dv)))
            e_info.append(np.float32(1.0 / np.float32(sig2[lvl])))
        pts.append(X)
        l += 1
    pts = np.array(pts)
    poses = np.zeros((P, 7))
    fixed = np.zeros(P, np.uint8)
    fixed[:n_fixed] = 1
    for i in range(P):
        R, t = Rs[i], ts[i]
        if not fixed[i]:
            R = _rot(rng.standard_normal(3) * pose_sigma) @ R
            t = t + rng.standard_normal(3) * pose_sigma
        # the reference feeds float32 4x4 poses through Converter::toSE3Quat (Converter.cc:37-47)
        R32 = R.astype(np.float32).astype(np.float64)
        poses[i, :4] = rot_to_quat(R32)
        poses[i, 4:] = t.astype(np.float32).astype(np.float64)
    pts_noisy = (pts + rng.standard_normal(pts.shape) * point_sigma).astype(np.float32).astype(np.float64)
    return dict(
        poses=poses, fixed=fixed, points=pts_noisy,
        e_point=np.array(e_point, np.int32), e_pose=np.array(e_pose, np.int32),
        e_obs=np.array(e_obs, np.float64).reshape(-1, 2), e_info=np.array(e_info, np.float64),
        K=np.array(KITTI_K, np.float64),
    )


def pose_problem(n=300, seed=2, outlier_frac=0.1, pose_sigma=0.02):
    """Pose-only problem for asd_pose_optimize (reference Optimizer.cc:239-413)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = KITTI_K
    R = _rot(rng.standard_normal(3) * 0.05)
    t = rng.standard_normal(3) * 0.3
    Xw, obs, info = [], [], []
    sig2 = 1.2 ** (2 * np.arange(8))
    while len(Xw) < n:
        Xc = np.array([rng.uniform(-12, 12), rng.uniform(-3, 3), rng.uniform(4, 40)])
        u = fx * Xc[0] / Xc[2] + cx
        v = fy * Xc[1] / Xc[2] + cy
        if not (0 <= u < KITTI_W and 0 <= v < KITTI_H):
            continue
        X = R.T @ (Xc - t)
        du, dv = rng.uniform(-0.5, 0.5, 2)
        if rng.uniform() < outlier_frac:
            du, dv = rng.uniform(-30, 30, 2)
        lvl = int(rng.integers(0, 8))
        Xw.append(X.astype(np.float32))
        obs.append((np.float32(u + du), np.float32(v + dv)))
        info.append(np.float32(1.0 / np.float32(sig2[lvl])))
    Rn = _rot(rng.standard_normal(3) * pose_sigma) @ R
    tn = t + rng.standard_normal(3) * pose_sigma * 5
    pose = np.zeros(7)
    pose[:4] = rot_to_quat(Rn.astype(np.float32).astype(np.float64))
    pose[4:] = tn.astype(np.float32).astype(np.float64)
    return dict(pose=pose, Xw=np.array(Xw, np.float64), obs=np.array(obs, np.float64),
                info=np.array(info, np.float64), K=np.array(KITTI_K, np.float64))


def vocabulary(k=10, L=3, seed=0, stop_frac=0.05, ragged=False):
    """Synthetic DBoW2-style vocabulary tree for 128-D unit descriptors (the reference's vocabulary file is not
    part of its tree): node descriptor = normalise(parent + noise whose norm halves per level), idf-like leaf
    weights with a few stopped words (weight 0).  Node ids are assigned breadth-first, children lists in creation
    order.  ragged=True prunes random subtrees so that leaves sit at different depths and branching varies.
    Returns a dict of the flat arrays asd_voc_load / orc_voc_create take."""
    rng = np.random.default_rng(seed)

    def spread(lv):
        return np.float32(1.0 if lv == 1 else 0.7 * 0.5 ** (lv - 2) / 11.3)

    if not ragged:  # level-at-a-time (a k=10, L=6 tree has 1.1 M nodes)
        descs = [np.zeros((1, 128), np.float32)]
        counts = [1]
        for lv in range(1, L + 1):
            par = descs[-1]
            d = par[:, None, :] + rng.standard_normal((len(par), k, 128), dtype=np.float32) * spread(lv)
            d /= np.linalg.norm(d, axis=2, keepdims=True)
            descs.append(d.reshape(-1, 128).astype(np.float32))
            counts.append(len(par) * k)
        n = int(sum(counts))
        desc = np.concatenate(descs)
        level = np.repeat(np.arange(L + 1, dtype=np.int32), counts)
        n_inner = n - counts[-1]
        child_start = np.concatenate([np.arange(n_inner + 1) * k, np.full(counts[-1], n_inner * k)]).astype(np.int32)
        child_ids = np.arange(1, n, dtype=np.int32)
        leaves = np.arange(n_inner, n)
    else:
        desc_l = [np.zeros(128, np.float32)]
        children = [[]]
        level_l = [0]
        frontier = [0]
        for lv in range(1, L + 1):
            nxt = []
            for p in frontier:
                nk = int(rng.integers(2, k + 1))
                if lv > 1 and rng.uniform() < 0.15:
                    continue  # p stays a leaf
                for _ in range(nk):
                    d = desc_l[p] + rng.standard_normal(128).astype(np.float32) * spread(lv)
                    d /= np.linalg.norm(d)
                    desc_l.append(d.astype(np.float32))
                    children.append([])
                    level_l.append(lv)
                    children[p].append(len(desc_l) - 1)
                    nxt.append(len(desc_l) - 1)
            frontier = nxt
        n = len(desc_l)
        desc = np.stack(desc_l)
        level = np.array(level_l, np.int32)
        child_start = np.zeros(n + 1, np.int32)
        for i in range(n):
            child_start[i + 1] = child_start[i] + len(children[i])
        child_ids = np.array([c for ch in children for c in ch], np.int32)
        leaves = np.array([i for i in range(1, n) if not children[i]])
    word_id = np.full(n, -1, np.int32)
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    weight = np.zeros(n, np.float64)
    weight[leaves] = rng.uniform(0.5, 9.0, len(leaves))
    weight[leaves[rng.uniform(size=len(leaves)) < stop_frac]] = 0.0
    return dict(n_nodes=n, k=k, L=L, child_start=child_start, child_ids=child_ids, weight=weight, word_id=word_id,
                desc=np.ascontiguousarray(desc, dtype=np.float32), level=level)
