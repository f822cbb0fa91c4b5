"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class orc_ba_out(C.Structure):
    _fields_ = [("chi2_first", C.c_double), ("chi2_second", C.c_double), ("iters_first", C.c_int32),
                ("iters_second", C.c_int32)]


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        for name, rt in (("orc_extractor_create", C.c_void_p), ("orc_frame_create", C.c_void_p),
                         ("orc_descriptor_distance", C.c_float), ("orc_fast_atan2", C.c_float),
                         ("orc_ic_angle", C.c_float)):
            if hasattr(L, name):
                getattr(L, name).restype = rt

    # ---- ASDNet
    def asdnet_forward(self, layers, patches, eps=1e-5, want_l6=False):
        ws = [_c(w, np.float32) for w, _, _ in layers]
        ms = [_c(m, np.float32) for _, m, _ in layers]
        vs = [_c(v, np.float32) for _, _, v in layers]
        arr = lambda xs: (C.c_void_p * 7)(*[x.ctypes.data for x in xs])
        patches = _c(patches, np.uint8).reshape(-1, 32, 32)
        n = patches.shape[0]
        out = np.empty((n, 128), np.float32)
        l6 = np.empty((n, 128, 8, 8), np.float32) if want_l6 else None
        self.lib.orc_asdnet_forward(arr(ws), arr(ms), arr(vs), C.c_float(eps), _p(patches), n, _p(out), _p(l6))
        return (out, l6) if want_l6 else out

    # ---- extractor
    def extractor(self, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=7):
        return OracleExtractor(self, nfeatures, scale, nlevels, ini_th, min_th)

    def resize_linear(self, src, dw, dh):
        src = _c(src, np.uint8)
        dst = np.empty((dh, dw), np.uint8)
        self.lib.orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
        return dst

    def gaussian_blur7(self, src):
        src = _c(src, np.uint8)
        dst = np.empty_like(src)
        self.lib.orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), src.shape[1])
        return dst

    def fast_score(self, img, x, y, th):
        img = _c(img, np.uint8)
        return self.lib.orc_fast_score(_p(img), img.strides[0], x, y, th)

    def fast_detect(self, img, th, cap=100000):
        img = _c(img, np.uint8)
        xs, ys, sc = (np.empty(cap, np.int32) for _ in range(3))
        n = self.lib.orc_fast_detect(_p(img), img.shape[1], img.shape[0], img.strides[0], th, cap, _p(xs), _p(ys),
                                     _p(sc))
        return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()

    def fast_atan2(self, y, x):
        return self.lib.orc_fast_atan2(C.c_float(y), C.c_float(x))

    # ---- frames / matchers
    def frame(self, kps, desc, bounds, nlevels=8, scale=1.2):
        return OracleFrame(self, kps, desc, bounds, nlevels, scale)

    def descriptor_distance(self, a, b):
        a, b = _c(a, np.float32), _c(b, np.float32)
        return self.lib.orc_descriptor_distance(_p(a), _p(b))

    def dist_matrix(self, a, b):
        a, b = _c(a, np.float32), _c(b, np.float32)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self.lib.orc_dist_matrix(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
        return out

    def bow_assemble(self, word, weight, node, weighting, scoring):
        return _bow_assemble(self.lib.orc_bow_assemble, word, weight, node, (int(weighting), int(scoring)))

    def distinctive_descriptor(self, desc):
        desc = _c(desc, np.float32)
        return self.lib.orc_distinctive_descriptor(_p(desc), desc.shape[0])

    def match_project_frame(self, cur, last, has_mp, Xw, mp_desc, Tcw, K, th, check_ori=True, obs_positive=None):
        has_mp, Xw, mp_desc = _c(has_mp, np.uint8), _c(Xw, np.float32), _c(mp_desc, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        out = np.empty(cur.n, np.int32)
        n = self.lib.orc_match_project_frame(cur.h, last.h, _p(has_mp), _p(Xw), _p(mp_desc), _p(Tcw), _p(K),
                                             C.c_float(th), int(check_ori), _p(out),
                                             _p(None if obs_positive is None else _c(obs_positive, np.uint8)))
        return out, n

    def match_project_points(self, cur, in_view, proj, level, view_cos, desc, occupied, th, nn_ratio, obs_positive=None):
        in_view, proj, level = _c(in_view, np.uint8), _c(proj, np.float32), _c(level, np.int32)
        view_cos, desc, occupied = _c(view_cos, np.float32), _c(desc, np.float32), _c(occupied, np.uint8)
        out = np.empty(cur.n, np.int32)
        n = self.lib.orc_match_project_points(cur.h, len(in_view), _p(in_view), _p(proj), _p(level), _p(view_cos),
                                              _p(desc), _p(occupied), C.c_float(th), C.c_float(nn_ratio), _p(out),
                                              _p(None if obs_positive is None else _c(obs_positive, np.uint8)))
        return out, n

    def frustum(self, cur, Xw, normal, min_dist, max_dist, Tcw, K, cos_limit=0.5):
        Xw, normal = _c(Xw, np.float32), _c(normal, np.float32)
        min_dist, max_dist = _c(min_dist, np.float32), _c(max_dist, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        n = Xw.shape[0]
        in_view = np.zeros(n, np.uint8)
        proj = np.zeros((n, 2), np.float32)
        level = np.zeros(n, np.int32)
        vc = np.zeros(n, np.float32)
        self.lib.orc_frustum(cur.h, n, _p(Xw), _p(normal), _p(min_dist), _p(max_dist), _p(Tcw), _p(K),
                             C.c_float(cos_limit), _p(in_view), _p(proj), _p(level), _p(vc))
        return in_view, proj, level, vc

    def match_init(self, f1, f2, prev_matched, window=100, nn_ratio=0.9, check_ori=True):
        pm = _c(prev_matched, np.float32).copy()
        out = np.empty(f1.n, np.int32)
        n = self.lib.orc_match_init(f1.h, f2.h, _p(pm), window, C.c_float(nn_ratio), int(check_ori), _p(out))
        return out, n, pm

    def fuse_search(self, kf, valid, Xw, normal, min_dist, max_dist, desc, Tcw, K, th=3.0):
        valid, Xw, normal = _c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32)
        min_dist, max_dist, desc = _c(min_dist, np.float32), _c(max_dist, np.float32), _c(desc, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        n = len(valid)
        bi = np.empty(n, np.int32)
        bd = np.empty(n, np.float32)
        self.lib.orc_fuse_search(kf.h, n, _p(valid), _p(Xw), _p(normal), _p(min_dist), _p(max_dist), _p(desc), _p(Tcw), _p(K),
                                 C.c_float(th), _p(bi), _p(bd))
        return bi, bd

    def svd4_vt(self, A):
        A = _c(A, np.float32).reshape(-1, 16)
        vt = np.empty((len(A), 4, 4), np.float32)
        for i in range(len(A)):
            self.lib.orc_svd4_vt(_p(A[i]), _p(vt[i]))
        return vt

    def triangulate_pairs(self, kps1, kps2, idx1, idx2, Tcw1, Tcw2, K1, K2, scale_factor=1.2, nlevels=8):
        idx1, idx2 = _c(idx1, np.int32), _c(idx2, np.int32)
        Tcw1, Tcw2, K1, K2 = _c(Tcw1, np.float32), _c(Tcw2, np.float32), _c(K1, np.float32), _c(K2, np.float32)
        k1, k2 = np.ascontiguousarray(kps1), np.ascontiguousarray(kps2)
        n = len(idx1)
        x = np.empty((n, 3), np.float32)
        ok = np.empty(n, np.uint8)
        self.lib.orc_triangulate_pairs.restype = C.c_int
        nok = self.lib.orc_triangulate_pairs(_p(k1), _p(k2), n, _p(idx1), _p(idx2), _p(Tcw1), _p(Tcw2), _p(K1), _p(K2),
                                             C.c_float(scale_factor), nlevels, _p(x), _p(ok))
        return x, ok, nok

    def match_project_keyframe(self, cur, valid, Xw, min_dist, max_dist, desc, kf_angle, occupied, Tcw, K, th, orb_dist, check_ori=True):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32), _c(desc, np.float32),
             _c(kf_angle, np.float32), _c(occupied, np.uint8), _c(Tcw, np.float32), _c(K, np.float32)]
        out = np.empty(cur.n, np.int32)
        self.lib.orc_match_project_keyframe.restype = C.c_int
        n = self.lib.orc_match_project_keyframe(cur.h, len(a[0]), *[_p(x) for x in a], C.c_float(th), C.c_float(orb_dist), int(check_ori), _p(out))
        return out, n

    def match_project_sim3(self, kf, Scw, valid, Xw, normal, min_dist, max_dist, desc, K, th, matched_kp):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32),
             _c(desc, np.float32), _c(K, np.float32)]
        Scw = _c(Scw, np.float32)
        mk = _c(matched_kp, np.int32).copy()
        self.lib.orc_match_project_sim3.restype = C.c_int
        n = self.lib.orc_match_project_sim3(kf.h, _p(Scw), len(a[0]), *[_p(x) for x in a], int(th), _p(mk))
        return mk, n

    def fuse_search_sim3(self, kf, Scw, valid, Xw, normal, min_dist, max_dist, desc, K, th=3.0):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32),
             _c(desc, np.float32), _c(K, np.float32)]
        Scw = _c(Scw, np.float32)
        bi, bd = np.empty(len(a[0]), np.int32), np.empty(len(a[0]), np.float32)
        self.lib.orc_fuse_search_sim3(kf.h, _p(Scw), len(a[0]), *[_p(x) for x in a], C.c_float(th), _p(bi), _p(bd))
        return bi, bd

    def match_sim3(self, kf1, kf2, has1, has2, Xw1, Xw2, mind1, maxd1, mind2, maxd2, desc1, desc2, T1w, T2w, s12, R12, t12, K, th):
        a = [_c(has1, np.uint8), _c(has2, np.uint8)] + [_c(x, np.float32) for x in (Xw1, Xw2, mind1, maxd1, mind2, maxd2, desc1, desc2, T1w, T2w)]
        R12, t12, K = _c(R12, np.float32), _c(t12, np.float32), _c(K, np.float32)
        out = np.empty(kf1.n, np.int32)
        self.lib.orc_match_sim3.restype = C.c_int
        n = self.lib.orc_match_sim3(kf1.h, kf2.h, *[_p(x) for x in a], C.c_float(s12), _p(R12), _p(t12), _p(K), C.c_float(th), _p(out))
        return out, n

    def vocabulary(self, voc, weighting=0, scoring=0):
        return OracleVocabulary(self.lib, voc, weighting, scoring)

    @staticmethod
    def _fv(node_of_kp):
        node_of_kp = np.asarray(node_of_kp)
        ids = np.unique(node_of_kp[node_of_kp >= 0]).astype(np.int32)
        start = np.zeros(len(ids) + 1, np.int32)
        idx = []
        for k, nid in enumerate(ids):
            members = np.nonzero(node_of_kp == nid)[0]
            idx.append(members)
            start[k + 1] = start[k] + len(members)
        idx = (np.concatenate(idx) if idx else np.zeros(0)).astype(np.int32)
        return ids, start, idx

    def match_bow(self, kf, f, nodes_kf, nodes_f, has_mp_kf, nn_ratio=0.7, check_ori=True):
        a, b = self._fv(nodes_kf), self._fv(nodes_f)
        has = _c(has_mp_kf, np.uint8)
        out = np.empty(f.n, np.int32)
        n = self.lib.orc_match_bow(kf.h, f.h, len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), len(b[0]), _p(b[0]), _p(b[1]),
                                   _p(b[2]), _p(has), C.c_float(nn_ratio), int(check_ori), _p(out))
        return out, n

    def match_triangulate(self, kf1, kf2, nodes1, nodes2, has_mp1, has_mp2, F12, ex, ey, check_ori=False):
        a, b = self._fv(nodes1), self._fv(nodes2)
        h1, h2, F = _c(has_mp1, np.uint8), _c(has_mp2, np.uint8), _c(F12, np.float32)
        out = np.empty(kf1.n, np.int32)
        n = self.lib.orc_match_triangulate(kf1.h, kf2.h, len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), len(b[0]), _p(b[0]),
                                           _p(b[1]), _p(b[2]), _p(h1), _p(h2), _p(F), C.c_float(ex), C.c_float(ey),
                                           int(check_ori), _p(out))
        return out, n

    # ---- optimizer
    def pose_optimize(self, pose7, Xw, obs, inv_sigma2, K):
        pose = _c(pose7, np.float64).copy()
        Xw, obs, inv_sigma2, K = (_c(a, np.float64) for a in (Xw, obs, inv_sigma2, K))
        outlier = np.zeros(Xw.shape[0], np.uint8)
        ninl = self.lib.orc_pose_optimize(_p(pose), Xw.shape[0], _p(Xw), _p(obs), _p(inv_sigma2), _p(K), _p(outlier))
        return pose, outlier, ninl

    def local_ba(self, prob, its_first=5, its_second=10):
        poses = _c(prob["poses"], np.float64).copy()
        points = _c(prob["points"], np.float64).copy()
        fixed = _c(prob["fixed"], np.uint8)
        e_point, e_pose = _c(prob["e_point"], np.int32), _c(prob["e_pose"], np.int32)
        e_obs, e_info, K = _c(prob["e_obs"], np.float64), _c(prob["e_info"], np.float64), _c(prob["K"], np.float64)
        E = len(e_point)
        chi2 = np.zeros(E, np.float64)
        dpos = np.zeros(E, np.uint8)
        out1 = np.zeros(E, np.uint8)
        o = orc_ba_out()
        rc = self.lib.orc_local_ba(len(poses), len(points), E, _p(poses), _p(fixed), _p(points), _p(e_point),
                                   _p(e_pose), _p(e_obs), _p(e_info), _p(K), its_first, its_second, _p(chi2),
                                   _p(dpos), _p(out1), C.byref(o))
        assert rc == 0
        return dict(poses=poses, points=points, edge_chi2=chi2, edge_depth_pos=dpos, edge_outlier1=out1,
                    chi2_first=o.chi2_first, chi2_second=o.chi2_second, iters_first=o.iters_first,
                    iters_second=o.iters_second)

    def tcw_to_pose7(self, T):
        T = _c(T, np.float32)
        p = np.zeros(7, np.float64)
        self.lib.orc_tcw_to_pose7(_p(T), _p(p))
        return p

    def pose7_to_tcw(self, p):
        p = _c(p, np.float64)
        T = np.zeros((4, 4), np.float32)
        self.lib.orc_pose7_to_tcw(_p(p), _p(T))
        return T


def _bow_assemble(fn, word, weight, node, extra):
    word, node, weight = _c(word, np.int32), _c(node, np.int32), _c(weight, np.float64)
    n = len(word)
    bid, bval = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float64)
    fnode, fstart, fidx = np.empty(max(n, 1), np.int32), np.empty(n + 1, np.int32), np.empty(max(n, 1), np.int32)
    nn = C.c_int32()
    fn.restype = C.c_int
    nw = fn(n, _p(word), _p(weight), _p(node), *extra, _p(bid), _p(bval), _p(fnode), _p(fstart), _p(fidx), C.byref(nn))
    return (bid[:nw].copy(), bval[:nw].copy()), (fnode[:nn.value].copy(), fstart[:nn.value + 1].copy(), fidx[:fstart[nn.value]].copy())


class RefDBoW2:
    """The reference's own DBoW2::BowVector / FeatureVector compiled in place (oracle/ref_dbow2 -> oracle/_ref/libdbow2_ref.so):
    the assembly half of TemplatedVocabulary::transform on per-feature (word, weight, node) triples."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libdbow2_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libdbow2_ref.so"))

    def assemble(self, word, weight, node, weighting, scoring):
        # ScoringObject.h:72-89: every scoring normalises with L1 except L2Scoring (L2) and DotProductScoring (none)
        return _bow_assemble(self.lib.ref_bow_assemble, word, weight, node, (int(weighting), int(scoring != 5), int(scoring == 1)))


class RefG2O:
    """The reference's own vendored g2o compiled in place (oracle/ref_g2o -> oracle/_ref/libg2o_ref.so).
    Same calling convention as Oracle.pose_optimize / Oracle.local_ba."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libg2o_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libg2o_ref.so"))

    def pose_optimize(self, pose7, Xw, obs, inv_sigma2, K):
        pose = _c(pose7, np.float64).copy()
        Xw, obs, inv_sigma2, K = (_c(a, np.float64) for a in (Xw, obs, inv_sigma2, K))
        outlier = np.zeros(Xw.shape[0], np.uint8)
        ninl = self.lib.ref_pose_optimize(_p(pose), Xw.shape[0], _p(Xw), _p(obs), _p(inv_sigma2), _p(K), _p(outlier))
        return pose, outlier, ninl

    def local_ba(self, prob, its_first=5, its_second=10):
        poses = _c(prob["poses"], np.float64).copy()
        points = _c(prob["points"], np.float64).copy()
        fixed = _c(prob["fixed"], np.uint8)
        e_point, e_pose = _c(prob["e_point"], np.int32), _c(prob["e_pose"], np.int32)
        e_obs, e_info, K = _c(prob["e_obs"], np.float64), _c(prob["e_info"], np.float64), _c(prob["K"], np.float64)
        E = len(e_point)
        chi2 = np.zeros(E, np.float64)
        dpos = np.zeros(E, np.uint8)
        out1 = np.zeros(E, np.uint8)
        o = orc_ba_out()
        rc = self.lib.ref_local_ba(len(poses), len(points), E, _p(poses), _p(fixed), _p(points), _p(e_point),
                                   _p(e_pose), _p(e_obs), _p(e_info), _p(K), its_first, its_second, _p(chi2),
                                   _p(dpos), _p(out1), C.byref(o))
        assert rc == 0
        return dict(poses=poses, points=points, edge_chi2=chi2, edge_depth_pos=dpos, edge_outlier1=out1,
                    chi2_first=o.chi2_first, chi2_second=o.chi2_second, iters_first=o.iters_first,
                    iters_second=o.iters_second)


class OracleExtractor:
    def __init__(self, orc, nfeatures, scale, nlevels, ini_th, min_th):
        self.orc, self.lib = orc, orc.lib
        self.nlevels = nlevels
        self.h = C.c_void_p(self.lib.orc_extractor_create(nfeatures, C.c_float(scale), nlevels, ini_th, min_th))

    def __del__(self):
        try:
            self.lib.orc_extractor_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        s, i, g, ig = (np.zeros(n, np.float32) for _ in range(4))
        f = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.lib.orc_extractor_tables(self.h, _p(s), _p(i), _p(g), _p(ig), _p(f), _p(um))
        return dict(scale=s, inv_scale=i, sigma2=g, inv_sigma2=ig, features_per_level=f, umax=um)

    def extract(self, image, cap=20000, want_patches=True):
        image = _c(image, np.uint8)
        kps = np.zeros(cap, KP_DTYPE)
        patches = np.empty((cap, 32, 32), np.uint8) if want_patches else None
        n = self.lib.orc_extract_keypoints(self.h, _p(image), image.shape[1], image.shape[0], image.strides[0],
                                           _p(kps), _p(patches), cap)
        return kps[:n].copy(), (patches[:n].copy() if want_patches else None)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        self.lib.orc_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, level, blurred=False):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        self.lib.orc_level_image(self.h, level, int(blurred), _p(out))
        return out

    def stereo_match(self, right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
        """Frame::ComputeStereoMatches on the pyramids self (left) and `right` hold after their last extract"""
        kl, kr = _c(kps_l, KP_DTYPE), _c(kps_r, KP_DTYPE)
        dl, dr = _c(desc_l, np.float32), _c(desc_r, np.float32)
        u, d = np.empty(len(kl), np.float32), np.empty(len(kl), np.float32)
        self.lib.orc_stereo_match.restype = C.c_int
        n = self.lib.orc_stereo_match(self.h, right.h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), C.c_float(mb), C.c_float(mbf),
                                      _p(u), _p(d))
        return u, d, n

    def raw_corners(self, level, cap=200000):
        x, y, r = (np.empty(cap, np.float32) for _ in range(3))
        n = self.lib.orc_raw_corners(self.h, level, cap, _p(x), _p(y), _p(r))
        return x[:n].copy(), y[:n].copy(), r[:n].copy()


class OracleFrame:
    def __init__(self, orc, kps, desc, bounds, nlevels, scale):
        self.lib = orc.lib
        kps = _c(kps, KP_DTYPE)
        desc = _c(desc, np.float32)
        self.n = len(kps)
        self.h = C.c_void_p(self.lib.orc_frame_create(_p(kps), _p(desc), self.n, C.c_float(bounds[0]),
                                                      C.c_float(bounds[1]), C.c_float(bounds[2]),
                                                      C.c_float(bounds[3]), nlevels, C.c_float(scale)))

    def __del__(self):
        try:
            self.lib.orc_frame_destroy(self.h)
        except Exception:
            pass

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1, cap=8192):
        out = np.empty(cap, np.int32)
        n = self.lib.orc_features_in_area(self.h, C.c_float(x), C.c_float(y), C.c_float(r), min_level, max_level,
                                          cap, _p(out))
        return out[:n].copy()


class OracleVocabulary:
    def __init__(self, lib, voc, weighting=0, scoring=0):
        self.lib = lib
        cs, ci = _c(voc["child_start"], np.int32), _c(voc["child_ids"], np.int32)
        w, wid, d = _c(voc["weight"], np.float64), _c(voc["word_id"], np.int32), _c(voc["desc"], np.float32)
        lib.orc_voc_create.restype = C.c_void_p
        self.h = C.c_void_p(lib.orc_voc_create(int(voc["n_nodes"]), int(voc["k"]), int(voc["L"]), weighting, scoring,
                                               _p(cs), _p(ci), _p(w), _p(wid), _p(d)))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_voc_destroy(self.h)
            self.h = None

    def descend(self, desc, levelsup=4):
        desc = _c(desc, np.float32)
        n = len(desc)
        word, node, weight = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        self.lib.orc_bow_descend(self.h, _p(desc), n, levelsup, _p(word), _p(node), _p(weight))
        return word, node, weight

    def transform(self, desc, levelsup=4):
        desc = _c(desc, np.float32)
        n = len(desc)
        bid, bval = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float64)
        fnode, fstart, fidx = np.empty(max(n, 1), np.int32), np.empty(n + 1, np.int32), np.empty(max(n, 1), np.int32)
        nn = C.c_int32(0)
        self.lib.orc_bow_transform.restype = C.c_int
        nw = self.lib.orc_bow_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), _p(fnode), _p(fstart), _p(fidx), C.byref(nn))
        return (bid[:nw].copy(), bval[:nw].copy()), (fnode[:nn.value].copy(), fstart[:nn.value + 1].copy(), fidx[:fstart[nn.value]].copy())
