"""ctypes binding of libasdhip's C ABI (include/asd_slam.h).

Plumbing for tests and bench only -- the same entry points a C++ adapter in the reference's
catkin workspace would call.  There is no fallback: if the shared library
is missing or no HIP device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # ASDHIP_LIB: tuning aid to A/B differently built libraries; the default is the in-tree build
    return os.environ.get("ASDHIP_LIB") or os.path.join(_HERE, "libasdhip.so")


class AsdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libasdhip error {code}: {msg}")
        self.code = code


class asd_config(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_patches", C.c_int32), ("device", C.c_int32)]


KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])


class asd_feature_vector(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("start", C.c_void_p), ("idx", C.c_void_p)]


def make_fv(node_of_kp):
    """node id per keypoint (-1 = none) -> (struct, keep-alive arrays) in DBoW2::FeatureVector order"""
    node_of_kp = np.asarray(node_of_kp)
    ids = np.unique(node_of_kp[node_of_kp >= 0]).astype(np.int32)
    start = np.zeros(len(ids) + 1, np.int32)
    idx = []
    for k, nid in enumerate(ids):
        members = np.nonzero(node_of_kp == nid)[0]
        idx.append(members)
        start[k + 1] = start[k] + len(members)
    idx = (np.concatenate(idx) if idx else np.zeros(0)).astype(np.int32)
    fv = asd_feature_vector(len(ids), ids.ctypes.data, start.ctypes.data, idx.ctypes.data)
    return fv, (ids, start, idx)


class asd_kf_neighbor(C.Structure):
    _fields_ = [("slot", C.c_int32), ("has_mp", C.c_void_p), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float),
                ("Tcw", C.c_float * 16), ("K", C.c_float * 4)]


class asd_fuse_call(C.Structure):
    _fields_ = [("slot_kf", C.c_int32), ("first", C.c_int32), ("n", C.c_int32), ("Tcw", C.c_float * 16), ("K", C.c_float * 4)]


class asd_track_frame_args(C.Structure):
    _fields_ = [("slot_cur", C.c_int32), ("slot_last", C.c_int32), ("has_mp", C.c_void_p), ("Xw_last", C.c_void_p), ("last_rows", C.c_void_p),
                ("last_cand", C.c_void_p), ("last_obs_positive", C.c_void_p), ("Tcw", C.c_void_p), ("th", C.c_float), ("check_orientation", C.c_int32),
                ("n_cand", C.c_int32), ("cand_rows", C.c_void_p), ("cand_obs_positive", C.c_void_p), ("viewing_cos_limit", C.c_float),
                ("th_local", C.c_float), ("nn_ratio", C.c_float), ("K", C.c_void_p), ("pose7", C.c_void_p), ("pose1", C.c_void_p),
                ("match1", C.c_void_p), ("n_matches1", C.c_void_p), ("outlier1", C.c_void_p), ("n_inliers1", C.c_void_p),
                ("match2", C.c_void_p), ("n_matches2", C.c_void_p), ("outlier2", C.c_void_p), ("n_inliers2", C.c_void_p)]


class asd_ba_problem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("poses", C.c_void_p), ("fixed", C.c_void_p), ("points", C.c_void_p),
                ("e_point", C.c_void_p), ("e_pose", C.c_void_p), ("e_obs", C.c_void_p), ("e_info", C.c_void_p),
                ("K", C.c_double * 4), ("its_first", C.c_int32), ("its_second", C.c_int32)]


class asd_ba_result(C.Structure):
    _fields_ = [("edge_chi2", C.c_void_p), ("edge_depth_pos", C.c_void_p), ("edge_outlier1", C.c_void_p),
                ("chi2_first", C.c_double), ("chi2_second", C.c_double),
                ("iters_first", C.c_int32), ("iters_second", C.c_int32)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def load_library():
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return C.CDLL(path)


class AsdHip:
    """One asd_ctx (= one HIP device + stream)."""

    def __init__(self, n_features=2000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7,
                 max_width=1241, max_height=376, max_patches=None, device=0):
        self.lib = load_library()
        L = self.lib
        L.asd_last_error.restype = C.c_char_p
        L.asd_version.restype = C.c_char_p
        L.asd_ctx_stream.restype = C.c_void_p
        self.cfg = asd_config(n_features, scale_factor, n_levels, ini_th, min_th, max_width, max_height,
                              max_patches if max_patches is not None else 2 * n_features, device)
        self.ctx = C.c_void_p()
        rc = L.asd_ctx_create(C.byref(self.cfg), C.byref(self.ctx))
        if rc != 0:
            raise AsdError(rc, "asd_ctx_create failed (no usable HIP device? there is no CPU fallback)")
        self.n_levels = n_levels

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx:
            self.lib.asd_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AsdError(rc, self.lib.asd_last_error(self.ctx).decode())

    def last_error(self):
        return self.lib.asd_last_error(self.ctx).decode()

    def calibration_note(self):
        self.lib.asd_calibration_note.restype = C.c_char_p
        return self.lib.asd_calibration_note(self.ctx).decode()

    # ---- tables
    def scale_tables(self):
        n = self.n_levels
        s, i, g, ig = (np.zeros(n, np.float32) for _ in range(4))
        f = np.zeros(n, np.int32)
        self._chk(self.lib.asd_get_scale_tables(self.ctx, _p(s), _p(i), _p(g), _p(ig), _p(f)))
        return dict(scale=s, inv_scale=i, sigma2=g, inv_sigma2=ig, features_per_level=f)

    # ---- ASDNet
    def load_weights(self, layers, eps=1e-5):
        ws = [_c(w, np.float32) for w, _, _ in layers]
        ms = [_c(m, np.float32) for _, m, _ in layers]
        vs = [_c(v, np.float32) for _, _, v in layers]
        arr = lambda xs: (C.c_void_p * 7)(*[x.ctypes.data for x in xs])
        self._keep = (ws, ms, vs)
        self._chk(self.lib.asd_load_weights(self.ctx, arr(ws), arr(ms), arr(vs), C.c_float(eps)))

    def describe(self, patches):
        patches = _c(patches, np.uint8).reshape(-1, 32, 32)
        n = patches.shape[0]
        out = np.empty((n, 128), np.float32)
        self._chk(self.lib.asd_describe(self.ctx, _p(patches), n, _p(out)))
        return out

    # ---- device utilities
    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.asd_device_alloc(self.ctx, C.c_uint64(nbytes), C.byref(p)))
        return p

    def host_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.asd_host_alloc(self.ctx, C.c_uint64(nbytes), C.byref(p)))
        return p

    def device_free(self, p):
        self._chk(self.lib.asd_device_free(self.ctx, p))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.asd_memcpy_h2d(self.ctx, dptr, _p(arr), C.c_uint64(arr.nbytes)))

    def d2h(self, arr, dptr):
        self._chk(self.lib.asd_memcpy_d2h(self.ctx, _p(arr), dptr, C.c_uint64(arr.nbytes)))

    def sync(self):
        self._chk(self.lib.asd_sync(self.ctx))

    def describe_timed(self, d_patches, n, d_desc, reps):
        ms = C.c_float()
        self._chk(self.lib.asd_describe_timed(self.ctx, d_patches, n, d_desc, reps, C.byref(ms)))
        return ms.value

    def debug_act6(self, n):
        """test aid: conv6's output of the most recent forward, first n patches, f32 [n, 64, 128]"""
        out = np.empty((n, 64, 128), np.float32)
        self.lib.asd_debug_act6.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.lib.asd_debug_act6(self.ctx, n, _p(out)))
        return out

    def describe_device(self, d_patches, n, d_desc):
        self._chk(self.lib.asd_describe_device(self.ctx, d_patches, n, d_desc))

    def last_stage_ms(self, stage):
        ms = C.c_float()
        self._chk(self.lib.asd_last_stage_ms(self.ctx, stage.encode(), C.byref(ms)))
        return ms.value

    # ---- extractor
    def extract(self, image, n_features_override=0):
        image = _c(image, np.uint8)
        h, w = image.shape
        cap = self.cfg.max_patches
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.empty((cap, 128), np.float32)
        n = C.c_int32()
        self._chk(self.lib.asd_extract(self.ctx, _p(image), w, h, image.strides[0], n_features_override,
                                       _p(kps), _p(desc), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_device(self, d_image, w, h, stride, n_features_override=0):
        cap = self.cfg.max_patches
        if not hasattr(self, "_kps_buf"):
            self._kps_buf = np.zeros(cap, KP_DTYPE)
            self._desc_buf = np.empty((cap, 128), np.float32)
        n = C.c_int32()
        self._chk(self.lib.asd_extract_device(self.ctx, d_image, w, h, stride, n_features_override,
                                              _p(self._kps_buf), _p(self._desc_buf), C.byref(n)))
        return self._kps_buf[:n.value], self._desc_buf[:n.value]

    def extract_submit(self, image, w, h, stride, device_resident=True, n_features_override=0):
        self._chk(self.lib.asd_extract_submit(self.ctx, image, int(device_resident), w, h, stride, n_features_override))

    def extract_wait(self, view=False):
        """view=True: zero-copy numpy views of the library's result buffers (valid for two further submissions)"""
        n = C.c_int32()
        if view:
            pk, pd = C.c_void_p(), C.c_void_p()
            self._chk(self.lib.asd_extract_wait_view(self.ctx, C.byref(pk), C.byref(pd), C.byref(n)))
            if n.value == 0:
                return np.zeros(0, KP_DTYPE), np.zeros((0, 128), np.float32)
            kb = (C.c_char * (n.value * KP_DTYPE.itemsize)).from_address(pk.value)
            db = (C.c_float * (n.value * 128)).from_address(pd.value)
            return np.frombuffer(kb, dtype=KP_DTYPE, count=n.value), np.frombuffer(db, dtype=np.float32).reshape(n.value, 128)
        cap = self.cfg.max_patches
        if not hasattr(self, "_kps_buf"):
            self._kps_buf = np.zeros(cap, KP_DTYPE)
            self._desc_buf = np.empty((cap, 128), np.float32)
        self._chk(self.lib.asd_extract_wait(self.ctx, _p(self._kps_buf), _p(self._desc_buf), C.byref(n)))
        return self._kps_buf[:n.value], self._desc_buf[:n.value]

    def extract_last_view(self):
        self.lib.asd_extract_last_view.restype = C.c_uint64
        return int(self.lib.asd_extract_last_view(self.ctx))

    def extract_view_valid(self, view_id):
        self.lib.asd_extract_view_valid.restype = C.c_int32
        return bool(self.lib.asd_extract_view_valid(self.ctx, C.c_uint64(view_id)))

    def extract_hold(self, on=True):
        """asd_extract_hold: no further ASDNet forward of the read-ahead extractor is enqueued while on (a wait on a held submission ends it)"""
        self._chk(self.lib.asd_extract_hold(self.ctx, int(on)))

    def profile_enable(self, on=True):
        self._chk(self.lib.asd_profile_enable(self.ctx, int(on)))

    def profile_get(self, layer):
        ms, calls, patches = C.c_double(), C.c_int32(), C.c_int64()
        self._chk(self.lib.asd_profile_get(self.ctx, layer, C.byref(ms), C.byref(calls), C.byref(patches)))
        return ms.value, calls.value, patches.value

    def asdnet_split_mask(self):
        """Bit l = conv(l+2) runs on the split-operand kernel; 0 = every layer on the f32 MFMA kernels."""
        self.lib.asd_asdnet_split_mask.restype = C.c_int32
        return int(self.lib.asd_asdnet_split_mask(self.ctx))

    def asdnet_pieces(self):
        """2 = operands as two fp16 terms, three products (default); 3 = three bf16 terms, six products (ASD_ASDNET_MATH=bf16x3)."""
        self.lib.asd_asdnet_pieces.restype = C.c_int32
        return int(self.lib.asd_asdnet_pieces(self.ctx))

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self._chk(self.lib.asd_get_level_size(self.ctx, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def level_image(self, level, blurred=False):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        self._chk(self.lib.asd_get_level_image(self.ctx, level, int(blurred), _p(out)))
        return out

    def raw_corners(self, level, cap=200000):
        x, y, r = (np.empty(cap, np.float32) for _ in range(3))
        n = C.c_int32()
        self._chk(self.lib.asd_get_raw_corners(self.ctx, level, cap, _p(x), _p(y), _p(r), C.byref(n)))
        return x[:n.value].copy(), y[:n.value].copy(), r[:n.value].copy()

    # ---- frames / matchers
    def frame_set(self, slot, kps, desc, bounds):
        kps = _c(kps, KP_DTYPE)
        d = None if desc is None else _c(desc, np.float32)
        self._chk(self.lib.asd_frame_set(self.ctx, slot, _p(kps), _p(d), len(kps), C.c_float(bounds[0]),
                                         C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3])))

    def features_in_area(self, slot, x, y, r, min_level=-1, max_level=-1, cap=8192):
        out = np.empty(cap, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_frame_features_in_area(self.ctx, slot, C.c_float(x), C.c_float(y), C.c_float(r),
                                                      min_level, max_level, cap, _p(out), C.byref(n)))
        return out[:n.value].copy()

    def dist_matrix(self, a, b):
        a, b = _c(a, np.float32), _c(b, np.float32)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self._chk(self.lib.asd_dist_matrix(self.ctx, _p(a), a.shape[0], _p(b), b.shape[0], _p(out)))
        return out

    def distinctive_descriptor(self, desc):
        desc = _c(desc, np.float32)
        best = C.c_int32()
        self._chk(self.lib.asd_distinctive_descriptor(self.ctx, _p(desc), desc.shape[0], C.byref(best)))
        return best.value

    def match_project_frame(self, slot_cur, slot_last, n_cur, has_mp, Xw, mp_desc, Tcw, K, th, check_ori=True, obs_positive=None):
        has_mp, Xw, mp_desc = _c(has_mp, np.uint8), _c(Xw, np.float32), _c(mp_desc, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        out = np.empty(n_cur, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_frame(self.ctx, slot_cur, slot_last, _p(has_mp), _p(Xw), _p(mp_desc),
                                                   _p(Tcw), _p(K), C.c_float(th), int(check_ori), _p(out),
                                                   C.byref(n), _p(None if obs_positive is None else _c(obs_positive, np.uint8))))
        return out, n.value

    def match_project_points(self, slot_cur, n_cur, in_view, proj, level, view_cos, desc, occupied, th, nn_ratio, obs_positive=None):
        in_view, proj, level = _c(in_view, np.uint8), _c(proj, np.float32), _c(level, np.int32)
        view_cos, desc, occupied = _c(view_cos, np.float32), _c(desc, np.float32), _c(occupied, np.uint8)
        out = np.empty(n_cur, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_points(self.ctx, slot_cur, len(in_view), _p(in_view), _p(proj),
                                                    _p(level), _p(view_cos), _p(desc), _p(occupied), C.c_float(th),
                                                    C.c_float(nn_ratio), _p(out), C.byref(n),
                                                    _p(None if obs_positive is None else _c(obs_positive, np.uint8))))
        return out, n.value

    # ---- fused tracking chains (search + claim replay + PoseOptimization, one synchronisation)
    def track_motion_model(self, slot_cur, slot_last, n_cur, has_mp, Xw, mp_desc_or_rows, Tcw, K, th, pose7, check_ori=True, obs_positive=None, split=False):
        """mp_desc_or_rows: float [n][128] descriptors or int32 bank rows.  -> (match_cur, n_matches, pose7, outlier, n_inliers);
        split=True: asd_track_async + the call (returns None once the work is enqueued), track_finish() returns the tuple"""
        has_mp, Xw, Tcw, K = _c(has_mp, np.uint8), _c(Xw, np.float32), _c(Tcw, np.float32), _c(K, np.float32)
        d = np.asarray(mp_desc_or_rows)
        bank = d.dtype.kind in "iu"
        d = _c(d, np.int32 if bank else np.float32)
        match, outl = np.empty(n_cur, np.int32), np.empty(max(n_cur, 1), np.uint8)
        pose = _c(pose7, np.float64).copy()
        n, ninl = C.c_int32(), C.c_int32()
        fn = self.lib.asd_track_motion_model_bank if bank else self.lib.asd_track_motion_model
        if split:
            self._chk(self.lib.asd_track_async(self.ctx))
        self._chk(fn(self.ctx, slot_cur, slot_last, _p(has_mp), _p(Xw), _p(d), _p(Tcw), _p(K), C.c_float(th), int(check_ori),
                     _p(None if obs_positive is None else _c(obs_positive, np.uint8)), _p(pose), _p(match), C.byref(n), _p(outl), C.byref(ninl)))
        if split:
            self._track_job = (match, n, pose, outl, ninl, n_cur)   # the library writes these at track_finish
            return None
        return match, n.value, pose, outl[:n_cur], ninl.value

    def track_local_map(self, slot_cur, n_cur, in_view, proj, level, view_cos, desc_or_rows, mp_Xw, occupied, cur_Xw, th, nn_ratio, K, pose7,
                        obs_positive=None, split=False):
        in_view, proj, level = _c(in_view, np.uint8), _c(proj, np.float32), _c(level, np.int32)
        view_cos, mp_Xw, occupied, cur_Xw, K = _c(view_cos, np.float32), _c(mp_Xw, np.float32), _c(occupied, np.uint8), _c(cur_Xw, np.float32), _c(K, np.float32)
        d = np.asarray(desc_or_rows)
        bank = d.dtype.kind in "iu"
        d = _c(d, np.int32 if bank else np.float32)
        match, outl = np.empty(n_cur, np.int32), np.empty(max(n_cur, 1), np.uint8)
        pose = _c(pose7, np.float64).copy()
        n, ninl = C.c_int32(), C.c_int32()
        fn = self.lib.asd_track_local_map_bank if bank else self.lib.asd_track_local_map
        if split:
            self._chk(self.lib.asd_track_async(self.ctx))
        self._chk(fn(self.ctx, slot_cur, len(in_view), _p(in_view), _p(proj), _p(level), _p(view_cos), _p(d), _p(mp_Xw), _p(occupied), _p(cur_Xw),
                     C.c_float(th), C.c_float(nn_ratio), _p(None if obs_positive is None else _c(obs_positive, np.uint8)), _p(K), _p(pose),
                     _p(match), C.byref(n), _p(outl), C.byref(ninl)))
        if split:
            self._track_job = (match, n, pose, outl, ninl, n_cur)   # the library writes these at track_finish
            return None
        return match, n.value, pose, outl[:n_cur], ninl.value

    def track_local_points(self, slot_cur, n_cur, Xw, normal, min_dist, max_dist, desc_or_rows, Tcw, K, occupied, cur_Xw, th, nn_ratio, pose7,
                           cos_limit=0.5, obs_positive=None, split=False):
        """frustum test + windows + search + claims + PoseOptimization in one submission"""
        Xw, normal, min_dist, max_dist = (_c(a, np.float32) for a in (Xw, normal, min_dist, max_dist))
        Tcw, K, occupied, cur_Xw = _c(Tcw, np.float32), _c(K, np.float32), _c(occupied, np.uint8), _c(cur_Xw, np.float32)
        d = np.asarray(desc_or_rows)
        bank = d.dtype.kind in "iu"
        d = _c(d, np.int32 if bank else np.float32)
        match, outl = np.empty(n_cur, np.int32), np.empty(max(n_cur, 1), np.uint8)
        pose = _c(pose7, np.float64).copy()
        n, ninl = C.c_int32(), C.c_int32()
        fn = self.lib.asd_track_local_points_bank if bank else self.lib.asd_track_local_points
        if split:
            self._chk(self.lib.asd_track_async(self.ctx))
        self._chk(fn(self.ctx, slot_cur, len(min_dist), _p(Xw), _p(normal), _p(min_dist), _p(max_dist), _p(d), _p(Tcw), _p(K), C.c_float(cos_limit),
                     _p(occupied), _p(cur_Xw), C.c_float(th), C.c_float(nn_ratio), _p(None if obs_positive is None else _c(obs_positive, np.uint8)),
                     _p(pose), _p(match), C.byref(n), _p(outl), C.byref(ninl)))
        if split:
            self._track_job = (match, n, pose, outl, ninl, n_cur)   # the library writes these at track_finish
            return None
        return match, n.value, pose, outl[:n_cur], ninl.value

    def track_local_points_rows(self, slot_cur, n_cur, rows, Tcw, K, occupied, cur_Xw, th, nn_ratio, pose7, cos_limit=0.5, obs_positive=None, split=False):
        """asd_track_local_points_rows: the map points by row of the descriptor + attribute banks"""
        rows = _c(rows, np.int32)
        Tcw, K, occupied, cur_Xw = _c(Tcw, np.float32), _c(K, np.float32), _c(occupied, np.uint8), _c(cur_Xw, np.float32)
        match, outl = np.empty(n_cur, np.int32), np.empty(max(n_cur, 1), np.uint8)
        pose = _c(pose7, np.float64).copy()
        n, ninl = C.c_int32(), C.c_int32()
        if split:
            self._chk(self.lib.asd_track_async(self.ctx))
        self._chk(self.lib.asd_track_local_points_rows(self.ctx, slot_cur, len(rows), _p(rows), _p(Tcw), _p(K), C.c_float(cos_limit), _p(occupied), _p(cur_Xw),
                                                       C.c_float(th), C.c_float(nn_ratio), _p(None if obs_positive is None else _c(obs_positive, np.uint8)),
                                                       _p(pose), _p(match), C.byref(n), _p(outl), C.byref(ninl)))
        if split:
            self._track_job = (match, n, pose, outl, ninl, n_cur)
            return None
        return match, n.value, pose, outl[:n_cur], ninl.value

    def mpbank_put(self, first_row, Xw, normal, min_dist, max_dist):
        Xw, normal, min_dist, max_dist = (_c(a, np.float32) for a in (Xw, normal, min_dist, max_dist))
        self._chk(self.lib.asd_mpbank_put(self.ctx, first_row, len(min_dist), _p(Xw), _p(normal), _p(min_dist), _p(max_dist)))

    def prep_async(self, on):
        self._chk(self.lib.asd_prep_async(self.ctx, int(on)))

    def track_frame(self, slot_cur, slot_last, n_cur, has_mp, Xw_last, last_rows, last_cand, Tcw, K, th, pose7, cand_rows, th_local, nn_ratio,
                    cos_limit=0.5, check_ori=True, last_obs_positive=None, cand_obs_positive=None, split=False):
        """asd_track_frame: both tracking stages as one submission -> dict(match1, n1, pose1, outlier1, n_inl1, match2, n2, pose, outlier2, n_inl2);
        split=True: returns None once enqueued, track_frame_finish() returns the dict"""
        a = asd_track_frame_args()
        keep = dict(has=_c(has_mp, np.uint8), Xw=_c(Xw_last, np.float32), rows=_c(last_rows, np.int32),
                    lc=None if last_cand is None else _c(last_cand, np.int32),
                    o1=None if last_obs_positive is None else _c(last_obs_positive, np.uint8), T=_c(Tcw, np.float32),
                    cr=_c(cand_rows, np.int32), o2=None if cand_obs_positive is None else _c(cand_obs_positive, np.uint8), K=_c(K, np.float32),
                    pose=_c(pose7, np.float64).copy(), pose1=np.zeros(7, np.float64),
                    m1=np.empty(n_cur, np.int32), m2=np.empty(n_cur, np.int32), out1=np.zeros(max(n_cur, 1), np.uint8), out2=np.zeros(max(n_cur, 1), np.uint8),
                    n1=C.c_int32(), n2=C.c_int32(), i1=C.c_int32(), i2=C.c_int32())
        adr = lambda x: None if x is None else x.ctypes.data
        a.slot_cur, a.slot_last = slot_cur, slot_last
        a.has_mp, a.Xw_last, a.last_rows, a.last_cand, a.last_obs_positive = adr(keep["has"]), adr(keep["Xw"]), adr(keep["rows"]), adr(keep["lc"]), adr(keep["o1"])
        a.Tcw, a.th, a.check_orientation = adr(keep["T"]), th, int(check_ori)
        a.n_cand, a.cand_rows, a.cand_obs_positive = len(keep["cr"]), adr(keep["cr"]), adr(keep["o2"])
        a.viewing_cos_limit, a.th_local, a.nn_ratio, a.K = cos_limit, th_local, nn_ratio, adr(keep["K"])
        a.pose7, a.pose1 = adr(keep["pose"]), adr(keep["pose1"])
        a.match1, a.n_matches1, a.outlier1, a.n_inliers1 = adr(keep["m1"]), C.addressof(keep["n1"]), adr(keep["out1"]), C.addressof(keep["i1"])
        a.match2, a.n_matches2, a.outlier2, a.n_inliers2 = adr(keep["m2"]), C.addressof(keep["n2"]), adr(keep["out2"]), C.addressof(keep["i2"])
        if split:
            self._chk(self.lib.asd_track_async(self.ctx))
        self._chk(self.lib.asd_track_frame(self.ctx, C.byref(a)))
        self._frame_job = (keep, n_cur)
        return None if split else self._frame_result()

    def _frame_result(self):
        k, n_cur = self._frame_job
        self._frame_job = None
        return dict(match1=k["m1"], n1=k["n1"].value, pose1=k["pose1"], outlier1=k["out1"][:n_cur], n_inl1=k["i1"].value,
                    match2=k["m2"], n2=k["n2"].value, pose=k["pose"], outlier2=k["out2"][:n_cur], n_inl2=k["i2"].value)

    def track_frame_finish(self):
        self._chk(self.lib.asd_track_finish(self.ctx))
        return self._frame_result()

    def track_finish(self):
        """completes the asd_track_* call started with split=True -> (match_cur, n_matches, pose7, outlier, n_inliers)"""
        rc = self.lib.asd_track_finish(self.ctx)
        job, self._track_job = getattr(self, "_track_job", None), None
        self._chk(rc)
        match, n, pose, outl, ninl, n_cur = job
        return match, n.value, pose, outl[:n_cur], ninl.value

    def debug_level_sweep(self, lo, hi):
        n = C.c_int64()
        self.lib.asd_debug_level_sweep.restype = C.c_int32
        bad = self.lib.asd_debug_level_sweep(self.ctx, C.c_float(lo), C.c_float(hi), C.byref(n))
        return bad, n.value

    def match_project_keyframe(self, slot_cur, n_cur, valid, Xw, min_dist, max_dist, desc, kf_angle, occupied, Tcw, K, th, orb_dist,
                               check_ori=True):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32), _c(desc, np.float32),
             _c(kf_angle, np.float32), _c(occupied, np.uint8), _c(Tcw, np.float32), _c(K, np.float32)]
        out = np.empty(n_cur, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_keyframe(self.ctx, slot_cur, len(a[0]), *[_p(x) for x in a], C.c_float(th), C.c_float(orb_dist),
                                                      int(check_ori), _p(out), C.byref(n)))
        return out, n.value

    def match_project_sim3(self, slot_kf, Scw, valid, Xw, normal, min_dist, max_dist, desc, K, th, matched_kp):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32),
             _c(desc, np.float32), _c(K, np.float32)]
        Scw = _c(Scw, np.float32)
        mk = _c(matched_kp, np.int32).copy()
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_sim3(self.ctx, slot_kf, _p(Scw), len(a[0]), *[_p(x) for x in a], int(th), _p(mk), C.byref(n)))
        return mk, n.value

    def fuse_search_sim3(self, slot_kf, Scw, valid, Xw, normal, min_dist, max_dist, desc, K, th=3.0):
        a = [_c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32),
             _c(desc, np.float32), _c(K, np.float32)]
        Scw = _c(Scw, np.float32)
        bi, bd = np.empty(len(a[0]), np.int32), np.empty(len(a[0]), np.float32)
        self._chk(self.lib.asd_fuse_search_sim3(self.ctx, slot_kf, _p(Scw), len(a[0]), *[_p(x) for x in a], C.c_float(th), _p(bi), _p(bd)))
        return bi, bd

    def match_sim3(self, slot1, slot2, n1, has1, has2, Xw1, Xw2, mind1, maxd1, mind2, maxd2, desc1, desc2, T1w, T2w, s12, R12, t12, K, th):
        a = [_c(has1, np.uint8), _c(has2, np.uint8)] + [_c(x, np.float32) for x in (Xw1, Xw2, mind1, maxd1, mind2, maxd2, desc1, desc2, T1w, T2w)]
        R12, t12, K = _c(R12, np.float32), _c(t12, np.float32), _c(K, np.float32)
        out = np.empty(n1, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_sim3(self.ctx, slot1, slot2, *[_p(x) for x in a], C.c_float(s12), _p(R12), _p(t12), _p(K), C.c_float(th),
                                          _p(out), C.byref(n)))
        return out, n.value

    def stereo_match(self, right_ctx, slot_left, slot_right, n_left, mb, mbf):
        u, d = np.empty(n_left, np.float32), np.empty(n_left, np.float32)
        n = C.c_int32()
        self._chk(self.lib.asd_stereo_match(self.ctx, right_ctx.ctx, slot_left, slot_right, C.c_float(mb), C.c_float(mbf), _p(u), _p(d),
                                            C.byref(n)))
        return u, d, n.value

    def distinctive_descriptor_batch(self, set_start, desc):
        set_start, desc = _c(set_start, np.int32), _c(desc, np.float32)
        out = np.empty(len(set_start) - 1, np.int32)
        self._chk(self.lib.asd_distinctive_descriptor_batch(self.ctx, len(out), _p(set_start), _p(desc), _p(out)))
        return out

    def fuse_search(self, slot_kf, valid, Xw, normal, min_dist, max_dist, desc, Tcw, K, th=3.0):
        valid, Xw, normal = _c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32)
        min_dist, max_dist, desc = _c(min_dist, np.float32), _c(max_dist, np.float32), _c(desc, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        n = len(valid)
        bi = np.empty(n, np.int32)
        bd = np.empty(n, np.float32)
        self._chk(self.lib.asd_fuse_search(self.ctx, slot_kf, n, _p(valid), _p(Xw), _p(normal), _p(min_dist), _p(max_dist),
                                           _p(desc), _p(Tcw), _p(K), C.c_float(th), _p(bi), _p(bd)))
        return bi, bd

    # ---- the per-keyframe stage, batched
    def frame_set_bow(self, slot, nodes):
        fv, keep = make_fv(nodes)
        self._chk(self.lib.asd_frame_set_bow(self.ctx, slot, C.byref(fv)))

    def create_map_points_batch(self, slot_cur, n_cur, has_cur, Tcw_cur, K_cur, neighbours):
        """neighbours: list of dict(slot, has_mp, F12, ex, ey, Tcw, K) -> matches [nb][n_cur], n_matches [nb], x3D [nb][n_cur][3], ok [nb][n_cur]"""
        nb = (asd_kf_neighbor * len(neighbours))()
        keep = []
        for b, d in enumerate(neighbours):
            h = _c(d["has_mp"], np.uint8); keep.append(h)
            nb[b].slot, nb[b].has_mp = d["slot"], h.ctypes.data
            nb[b].F12 = (C.c_float * 9)(*[float(v) for v in np.asarray(d["F12"], np.float32).ravel()])
            nb[b].ex, nb[b].ey = float(d["ex"]), float(d["ey"])
            nb[b].Tcw = (C.c_float * 16)(*[float(v) for v in np.asarray(d["Tcw"], np.float32).ravel()])
            nb[b].K = (C.c_float * 4)(*[float(v) for v in np.asarray(d["K"], np.float32).ravel()])
        has_cur, Tcw_cur, K_cur = _c(has_cur, np.uint8), _c(Tcw_cur, np.float32), _c(K_cur, np.float32)
        B = len(neighbours)
        m = np.empty((B, n_cur), np.int32); nm = np.zeros(B, np.int32)
        x = np.empty((B, n_cur, 3), np.float32); ok = np.empty((B, n_cur), np.uint8)
        self._chk(self.lib.asd_create_map_points_batch(self.ctx, slot_cur, _p(has_cur), _p(Tcw_cur), _p(K_cur), B, nb, _p(m), _p(nm), _p(x), _p(ok)))
        return m, nm, x, ok

    def fuse_search_batch(self, calls, valid, Xw, normal, min_dist, max_dist, desc, th=3.0):
        """calls: list of dict(slot_kf, first, n, Tcw, K) over the shared map-point tables -> best_idx, best_dist [n_total];
        desc: float [n][128] descriptors or int32 rows of the descriptor bank"""
        cs = (asd_fuse_call * len(calls))()
        for c, d in enumerate(calls):
            cs[c].slot_kf, cs[c].first, cs[c].n = d["slot_kf"], d["first"], d["n"]
            cs[c].Tcw = (C.c_float * 16)(*[float(v) for v in np.asarray(d["Tcw"], np.float32).ravel()])
            cs[c].K = (C.c_float * 4)(*[float(v) for v in np.asarray(d["K"], np.float32).ravel()])
        valid, Xw, normal = _c(valid, np.uint8), _c(Xw, np.float32), _c(normal, np.float32)
        min_dist, max_dist = _c(min_dist, np.float32), _c(max_dist, np.float32)
        d = np.asarray(desc)
        rows = d.dtype.kind in "iu"
        d = _c(d, np.int32 if rows else np.float32)
        n = len(valid)
        bi, bd = np.empty(n, np.int32), np.empty(n, np.float32)
        self._chk(self.lib.asd_fuse_search_batch(self.ctx, len(calls), cs, n, _p(valid), _p(Xw), _p(normal), _p(min_dist), _p(max_dist),
                                                 _p(None if rows else d), _p(d if rows else None), C.c_float(th), _p(bi), _p(bd)))
        return bi, bd

    def triangulate_pairs(self, slot1, slot2, idx1, idx2, Tcw1, Tcw2, K1, K2):
        idx1, idx2 = _c(idx1, np.int32), _c(idx2, np.int32)
        Tcw1, Tcw2, K1, K2 = _c(Tcw1, np.float32), _c(Tcw2, np.float32), _c(K1, np.float32), _c(K2, np.float32)
        n = len(idx1)
        x = np.empty((n, 3), np.float32)
        ok = np.empty(n, np.uint8)
        nok = C.c_int32(0)
        self._chk(self.lib.asd_triangulate_pairs(self.ctx, slot1, slot2, n, _p(idx1), _p(idx2), _p(Tcw1), _p(Tcw2), _p(K1), _p(K2),
                                                 _p(x), _p(ok), C.byref(nok)))
        return x, ok, nok.value

    def svd4_null(self, A):
        A = _c(A, np.float32).reshape(-1, 16)
        v = np.empty((len(A), 4), np.float32)
        self._chk(self.lib.asd_svd4_null(self.ctx, len(A), _p(A), _p(v)))
        return v

    # ---- vocabulary / BoW
    def voc_load(self, voc, weighting=0, scoring=0):
        cs, ci = _c(voc["child_start"], np.int32), _c(voc["child_ids"], np.int32)
        w, wid, d = _c(voc["weight"], np.float64), _c(voc["word_id"], np.int32), _c(voc["desc"], np.float32)
        self._chk(self.lib.asd_voc_load(self.ctx, int(voc["n_nodes"]), int(voc["k"]), int(voc["L"]), weighting, scoring,
                                        _p(cs), _p(ci), _p(w), _p(wid), _p(d)))

    def bow_descend(self, desc=None, slot=-1, n=None, levelsup=4):
        if desc is not None:
            desc = _c(desc, np.float32)
            n = len(desc)
        word, node, weight = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        self._chk(self.lib.asd_bow_descend(self.ctx, slot, _p(desc) if desc is not None else None, n, levelsup,
                                           _p(word), _p(node), _p(weight)))
        return word, node, weight

    def compute_bow(self, desc=None, slot=-1, n=None, levelsup=4):
        """-> (bow_id, bow_val), (fv_node, fv_start, fv_idx)"""
        if desc is not None:
            desc = _c(desc, np.float32)
            n = len(desc)
        bid, bval = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float64)
        fnode, fstart, fidx = np.empty(max(n, 1), np.int32), np.empty(n + 1, np.int32), np.empty(max(n, 1), np.int32)
        nw, nn = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.asd_compute_bow(self.ctx, slot, _p(desc) if desc is not None else None, n, levelsup, _p(bid), _p(bval),
                                           C.byref(nw), _p(fnode), _p(fstart), _p(fidx), C.byref(nn)))
        return (bid[:nw.value].copy(), bval[:nw.value].copy()), \
               (fnode[:nn.value].copy(), fstart[:nn.value + 1].copy(), fidx[:fstart[nn.value]].copy())

    def bank_put(self, first_row, desc):
        desc = _c(desc, np.float32)
        self._chk(self.lib.asd_bank_put(self.ctx, first_row, desc.shape[0], _p(desc)))

    def bank_put_from_frame(self, slot, first_row, n):
        self._chk(self.lib.asd_bank_put_from_frame(self.ctx, slot, first_row, n))

    def match_project_frame_bank(self, slot_cur, slot_last, n_cur, has_mp, Xw, mp_rows, Tcw, K, th, check_ori=True, obs_positive=None):
        has_mp, Xw, mp_rows = _c(has_mp, np.uint8), _c(Xw, np.float32), _c(mp_rows, np.int32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        out = np.empty(n_cur, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_frame_bank(self.ctx, slot_cur, slot_last, _p(has_mp), _p(Xw), _p(mp_rows),
                                                        _p(Tcw), _p(K), C.c_float(th), int(check_ori), _p(out), C.byref(n),
                                                        _p(None if obs_positive is None else _c(obs_positive, np.uint8))))
        return out, n.value

    def match_project_points_bank(self, slot_cur, n_cur, in_view, proj, level, view_cos, rows, occupied, th, nn_ratio, obs_positive=None):
        in_view, proj, level = _c(in_view, np.uint8), _c(proj, np.float32), _c(level, np.int32)
        view_cos, rows, occupied = _c(view_cos, np.float32), _c(rows, np.int32), _c(occupied, np.uint8)
        out = np.empty(n_cur, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_project_points_bank(self.ctx, slot_cur, len(in_view), _p(in_view), _p(proj), _p(level),
                                                         _p(view_cos), _p(rows), _p(occupied), C.c_float(th),
                                                         C.c_float(nn_ratio), _p(out), C.byref(n),
                                                         _p(None if obs_positive is None else _c(obs_positive, np.uint8))))
        return out, n.value

    def frustum(self, slot_cur, Xw, normal, min_dist, max_dist, Tcw, K, cos_limit=0.5):
        Xw, normal = _c(Xw, np.float32), _c(normal, np.float32)
        min_dist, max_dist = _c(min_dist, np.float32), _c(max_dist, np.float32)
        Tcw, K = _c(Tcw, np.float32), _c(K, np.float32)
        n = Xw.shape[0]
        in_view = np.zeros(n, np.uint8)
        proj = np.zeros((n, 2), np.float32)
        level = np.zeros(n, np.int32)
        vc = np.zeros(n, np.float32)
        self._chk(self.lib.asd_frustum(self.ctx, slot_cur, n, _p(Xw), _p(normal), _p(min_dist), _p(max_dist),
                                       _p(Tcw), _p(K), C.c_float(cos_limit), _p(in_view), _p(proj), _p(level),
                                       _p(vc)))
        return in_view, proj, level, vc

    def match_init(self, slot1, slot2, prev_matched, window=100, nn_ratio=0.9, check_ori=True):
        pm = _c(prev_matched, np.float32).copy()
        out = np.empty(pm.shape[0], np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_init(self.ctx, slot1, slot2, _p(pm), window, C.c_float(nn_ratio),
                                          int(check_ori), _p(out), C.byref(n)))
        return out, n.value, pm

    def match_bow(self, slot_kf, slot_f, n_f, nodes_kf, nodes_f, has_mp_kf, nn_ratio=0.7, check_ori=True):
        fa, ka = make_fv(nodes_kf)
        fb, kb = make_fv(nodes_f)
        has = _c(has_mp_kf, np.uint8)
        out = np.empty(n_f, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_bow(self.ctx, slot_kf, slot_f, C.byref(fa), C.byref(fb), _p(has), C.c_float(nn_ratio),
                                         int(check_ori), _p(out), C.byref(n)))
        return out, n.value

    def match_triangulate(self, slot1, slot2, n1, nodes1, nodes2, has_mp1, has_mp2, F12, ex, ey, check_ori=False):
        fa, ka = make_fv(nodes1)
        fb, kb = make_fv(nodes2)
        h1, h2, F = _c(has_mp1, np.uint8), _c(has_mp2, np.uint8), _c(F12, np.float32)
        out = np.empty(n1, np.int32)
        n = C.c_int32()
        self._chk(self.lib.asd_match_triangulate(self.ctx, slot1, slot2, C.byref(fa), C.byref(fb), _p(h1), _p(h2), _p(F),
                                                 C.c_float(ex), C.c_float(ey), int(check_ori), _p(out), C.byref(n)))
        return out, n.value

    # ---- optimizer
    def pose_optimize(self, pose7, Xw, obs, inv_sigma2, K):
        pose = _c(pose7, np.float64).copy()
        Xw, obs, inv_sigma2, K = (_c(a, np.float64) for a in (Xw, obs, inv_sigma2, K))
        n = Xw.shape[0]
        outlier = np.zeros(n, np.uint8)
        ninl = C.c_int32()
        self._chk(self.lib.asd_pose_optimize(self.ctx, _p(pose), n, _p(Xw), _p(obs), _p(inv_sigma2), _p(K),
                                             _p(outlier), C.byref(ninl)))
        return pose, outlier, ninl.value

    def _ba_pack(self, prob, its_first, its_second):
        poses = _c(prob["poses"], np.float64).copy()
        points = _c(prob["points"], np.float64).copy()
        fixed = _c(prob["fixed"], np.uint8)
        e_point, e_pose = _c(prob["e_point"], np.int32), _c(prob["e_pose"], np.int32)
        e_obs, e_info = _c(prob["e_obs"], np.float64), _c(prob["e_info"], np.float64)
        E = len(e_point)
        chi2 = np.zeros(E, np.float64)
        dpos = np.zeros(E, np.uint8)
        out1 = np.zeros(E, np.uint8)
        p = asd_ba_problem(len(poses), len(points), E, poses.ctypes.data, fixed.ctypes.data, points.ctypes.data,
                           e_point.ctypes.data, e_pose.ctypes.data, e_obs.ctypes.data, e_info.ctypes.data,
                           (C.c_double * 4)(*[float(k) for k in prob["K"]]), its_first, its_second)
        r = asd_ba_result(chi2.ctypes.data, dpos.ctypes.data, out1.ctypes.data, 0.0, 0.0, 0, 0)
        keep = (poses, points, fixed, e_point, e_pose, e_obs, e_info, chi2, dpos, out1)   # the structs hold raw addresses
        return p, r, keep

    @staticmethod
    def _ba_unpack(r, keep):
        poses, points, _f, _ep, _es, _eo, _ei, chi2, dpos, out1 = keep
        return dict(poses=poses, points=points, edge_chi2=chi2, edge_depth_pos=dpos, edge_outlier1=out1,
                    chi2_first=r.chi2_first, chi2_second=r.chi2_second, iters_first=r.iters_first,
                    iters_second=r.iters_second)

    def local_ba(self, prob, its_first=5, its_second=10):
        p, r, keep = self._ba_pack(prob, its_first, its_second)
        self._chk(self.lib.asd_local_ba(self.ctx, C.byref(p), C.byref(r)))
        return self._ba_unpack(r, keep)

    def local_ba_submit(self, prob, its_first=5, its_second=10):
        """LocalBundleAdjustment on the library's OPTIONAL lane: returns at once; local_ba_wait() returns the result dict.  One run
        at a time.  Not the reference's order -- Tracking.cc:797 -> LocalMapping.cc:89 runs it in line (local_ba above); frames tracked
        while a run is out read the pre-BA map."""
        p, r, keep = self._ba_pack(prob, its_first, its_second)
        self._chk(self.lib.asd_local_ba_submit(self.ctx, C.byref(p), C.byref(r)))
        self._ba_job = (p, r, keep)   # the library owns these until wait

    def local_ba_wait(self):
        rc = self.lib.asd_local_ba_wait(self.ctx)
        job, self._ba_job = getattr(self, "_ba_job", None), None
        self._chk(rc)
        p, r, keep = job
        return self._ba_unpack(r, keep)

    def local_ba_poll(self):
        return int(self.lib.asd_local_ba_poll(self.ctx))

    def tcw_to_pose7(self, T):
        T = _c(T, np.float32)
        p = np.zeros(7, np.float64)
        self.lib.asd_tcw_to_pose7(_p(T), _p(p))
        return p

    def pose7_to_tcw(self, p):
        p = _c(p, np.float64)
        T = np.zeros((4, 4), np.float32)
        self.lib.asd_pose7_to_tcw(_p(p), _p(T))
        return T
