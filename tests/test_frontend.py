"""Extractor front-end: oracle known-answer tests (OpenCV 3.2.0 semantics are
parity-unpinned: no OpenCV here, no reference vectors) and HIP-vs-oracle bit-exactness."""
import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_fast_score(img, x, y, th):
    """FAST-9/16 segment test + score straight from the definition."""
    v = int(img[y, x])
    d = [v - int(img[y + dy, x + dx]) for dx, dy in RING]
    best = -999
    for k in range(16):
        arc = [d[(k + i) % 16] for i in range(9)]
        best = max(best, min(arc), min(-a for a in arc))
    return best - 1 if best > th else 0


# ------------------------------------------------------------------ oracle known answers (CPU)
def test_resize_constant_and_half(oracle):
    src = np.full((40, 60), 137, np.uint8)
    np.testing.assert_array_equal(oracle.resize_linear(src, 50, 33), 137)
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (40, 60)).astype(np.uint8)
    half = oracle.resize_linear(src, 30, 20)
    s = src.astype(np.int32)
    # exact 2:1: both coefficients are 1024 -> ((1024*(r>>4))>>16)*2 with r = 1024*(a+b)
    r0 = 1024 * (s[0::2, 0::2] + s[0::2, 1::2])
    r1 = 1024 * (s[1::2, 0::2] + s[1::2, 1::2])
    exp = (((1024 * (r0 >> 4)) >> 16) + ((1024 * (r1 >> 4)) >> 16) + 2) >> 2
    np.testing.assert_array_equal(half, exp.astype(np.uint8))
    # identity resize is the identity
    np.testing.assert_array_equal(oracle.resize_linear(src, 60, 40), src)


def test_blur_kernel_and_borders(oracle):
    k = np.array([18, 34, 49, 55, 49, 34, 18])  # round(256 * getGaussianKernel(7, 2)): sums to 257
    assert k.sum() == 257
    c = np.full((20, 30), 100, np.uint8)
    np.testing.assert_array_equal(oracle.gaussian_blur7(c), (100 * 257 * 257 + 32768) >> 16)
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = oracle.gaussian_blur7(imp)
    exp = (255 * np.outer(k, k) + 32768) >> 16
    np.testing.assert_array_equal(out[7:14, 7:14], exp)
    assert out[:7].sum() == 0 and out[14:].sum() == 0
    # reflect-101 (row -1 mirrors row 1): an impulse on row 1 is seen twice by rows 0..2
    imp = np.zeros((21, 21), np.uint8)
    imp[1, 10] = 255
    out = oracle.gaussian_blur7(imp)
    col = np.array([k[4] + k[2], k[3] + k[1], k[2] + k[0], k[1], k[0]])
    np.testing.assert_array_equal(out[:5, 10], (255 * col * k[3] + 32768) >> 16)


def test_fast_score_definition(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    img[10:30, 10:30] = np.clip(img[10:30, 10:30] // 8 + 100, 0, 255)
    for th in (7, 20):
        for y in range(3, 37):
            for x in range(3, 37):
                assert oracle.fast_score(img, x, y, th) == brute_fast_score(img, x, y, th), (x, y, th)
    # a bright dot on a dark background is a corner with score = contrast - 1
    dot = np.full((9, 9), 10, np.uint8)
    dot[4, 4] = 110
    assert oracle.fast_score(dot, 4, 4, 20) == 99


def test_fast_detect_nms(oracle):
    rng = np.random.default_rng(2)
    img = (rng.integers(0, 2, (36, 37)) * 120 + rng.integers(0, 40, (36, 37))).astype(np.uint8)
    xs, ys, sc = oracle.fast_detect(img, 20)
    S = np.zeros(img.shape, np.int32)
    for y in range(3, 33):
        for x in range(3, 34):
            S[y, x] = brute_fast_score(img, x, y, 20)
    exp = []
    for y in range(3, 33):
        for x in range(3, 34):
            s = S[y, x]
            if s > 0 and all(s > S[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx or dy)):
                exp.append((x, y, s))
    assert len(exp) > 5
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == exp


def test_fast_atan2(oracle):
    rng = np.random.default_rng(3)
    for y, x in rng.standard_normal((200, 2)) * 1000:
        a = oracle.fast_atan2(y, x)
        ref = np.degrees(np.arctan2(y, x)) % 360
        assert min(abs(a - ref), 360 - abs(a - ref)) < 0.3
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, -5.0) == 180.0


def test_oracle_extract_properties(oracle, synth):
    img = synth.scene_frame(0)
    ex = oracle.extractor(2000)
    t = ex.tables()
    assert t["features_per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert [ex_ for ex_ in t["umax"]] == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    kps, patches = ex.extract(img)
    sizes = [ex.level_size(l) for l in range(8)]
    assert sizes == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert 1900 <= len(kps) <= 2100
    assert (np.diff(kps["octave"]) >= 0).all()  # level-major output order
    for l in range(8):
        m = kps["octave"] == l
        w, h = sizes[l]
        x = kps["x"][m] / t["scale"][l]
        y = kps["y"][m] / t["scale"][l]
        assert x.min() >= 19 - 1e-3 and x.max() <= w - 20 + 1e-3 and y.min() >= 19 - 1e-3 and y.max() <= h - 20 + 1e-3
        assert m.sum() >= t["features_per_level"][l]  # the quadtree stops at >= N nodes
        # one keypoint per quadtree leaf -> unique positions
        assert len({(a, b) for a, b in zip(kps["x"][m].tolist(), kps["y"][m].tolist())}) == m.sum()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360.0001)).all()
    # patch = 32x32 of the blurred level with the keypoint at [16][16]
    l = int(kps["octave"][5])
    bl = ex.level_image(l, blurred=True)
    x = int(round(kps["x"][5] / t["scale"][l])); y = int(round(kps["y"][5] / t["scale"][l]))
    np.testing.assert_array_equal(patches[5], bl[y - 16:y + 16, x - 16:x + 16])


def test_oracle_extract_flat_image(oracle):
    ex = oracle.extractor(500)
    kps, _ = ex.extract(np.full((240, 320), 90, np.uint8))
    assert len(kps) == 0


# ------------------------------------------------------------------ HIP vs oracle (GPU)
def _compare_extract(hip, oracle, synth, img, nfeat, override=0):
    layers = synth.asdnet_weights(0)
    kps, desc = hip.extract(img, n_features_override=override)
    ex = oracle.extractor(nfeat)
    okps, opatches = ex.extract(img)
    for l in range(8):
        assert hip.level_size(l) == ex.level_size(l)
        np.testing.assert_array_equal(hip.level_image(l), ex.level_image(l), err_msg=f"pyramid level {l}")
        hx, hy, hr = hip.raw_corners(l)
        ox, oy, orr = ex.raw_corners(l)
        assert len(hx) == len(ox), f"raw corner count level {l}: {len(hx)} vs {len(ox)}"
        np.testing.assert_array_equal(hx, ox)
        np.testing.assert_array_equal(hy, oy)
        np.testing.assert_array_equal(hr, orr)
        if (okps["octave"] == l).any():
            np.testing.assert_array_equal(hip.level_image(l, blurred=True), ex.level_image(l, blurred=True),
                                          err_msg=f"blurred level {l}")
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        np.testing.assert_array_equal(kps[f], okps[f], err_msg=f)  # bit-exact keypoints
    # descriptors: the oracle's naive conv is slow, check every 8th keypoint (all levels are covered)
    sub = np.arange(0, len(kps), 8)
    odesc = oracle.asdnet_forward(layers, opatches[sub])
    np.testing.assert_allclose(desc[sub], odesc, atol=2e-5, rtol=0)
    return kps, desc


@pytest.mark.gpu
def test_extract_kitti_size_bit_exact(hip, oracle, synth):
    kps, desc = _compare_extract(hip, oracle, synth, synth.scene_frame(0), 2000)
    assert len(kps) >= 2000


@pytest.mark.gpu
def test_extract_other_frame_and_init_quota(hip, oracle, synth):
    # Tracking.cc:85: the initialisation extractor asks for 2 x nFeatures
    kps, _ = _compare_extract(hip, oracle, synth, synth.scene_frame(7), 4000, override=4000)
    assert len(kps) >= 3900


@pytest.mark.gpu
def test_extract_small_and_odd_sizes(hip, oracle, synth):
    img = synth.scene_frame(3)[:251, :333]
    _compare_extract(hip, oracle, synth, np.ascontiguousarray(img), 2000)
    # strided input (stride > width)
    big = synth.scene_frame(4)
    view = big[10:300, 100:900]
    kps_a, desc_a = hip.extract(np.ascontiguousarray(view))
    assert len(kps_a) > 500


@pytest.mark.gpu
def test_extract_flat_and_low_texture(hip, oracle, synth):
    kps, desc = hip.extract(np.full((376, 1241), 128, np.uint8))
    assert len(kps) == 0 and desc.shape == (0, 128)
    # low contrast: only the min-threshold retry (ORBextractor.cc:861-866) finds corners
    img = (synth.scene_frame(1).astype(np.int32) - 128) // 8 + 128
    _compare_extract(hip, oracle, synth, img.astype(np.uint8), 2000)


@pytest.mark.gpu
def test_extract_deterministic(hip, synth):
    img = synth.scene_frame(2)
    k1, d1 = hip.extract(img)
    k2, d2 = hip.extract(img)
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(d1, d2)


@pytest.mark.gpu
def test_synchronous_entry_points_refuse_to_run_under_outstanding_submissions(hip, synth):
    """The extraction worker owns the shared pyramid / score / activation buffers while a submission is outstanding:
    asd_extract, asd_describe and the debug read-backs return an error instead of racing with it, and work again once
    every submission has been waited for."""
    im = synth.scene_frame(3)
    ref = hip.extract(im)
    p = hip.device_alloc(im.nbytes)
    hip.h2d(p, im)
    hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
    with pytest.raises(Exception, match="outstanding"):
        hip.extract(im)
    with pytest.raises(Exception, match="outstanding"):
        hip.describe(synth.random_patches(8))
    with pytest.raises(Exception, match="outstanding"):
        hip.level_image(0)
    kps, desc = hip.extract_wait()
    np.testing.assert_array_equal(kps, ref[0])
    np.testing.assert_array_equal(desc, ref[1])
    again = hip.extract(im)
    np.testing.assert_array_equal(again[1], ref[1])
    hip.device_free(p)


@pytest.mark.gpu
def test_dist_matrix_beside_readahead_extractor(hip, synth):
    """Other kernels of the library keep returning exact values while the read-ahead extractor's bf16-MFMA kernels share
    the CUs with them.  On gfx950 a packed-f32 instruction with op_sel:[0,1] loses results in lanes 48-63 beside ANY bf16
    MFMA kernel (tools/ubench/mfma_pk_hazard.hip, profiles/r02_mfma_pk_hazard.txt): a library built
    with the SLP vectoriser fails this test in 23 of 24 calls, the shipped build (no packed-f32 arithmetic, tests/test_isa.py)
    must not fail it once."""
    im = synth.scene_frame(5)
    p = hip.device_alloc(im.nbytes)
    hip.h2d(p, im)
    n = 300
    inputs = [synth.unit_descriptors(n, seed=k) for k in range(12)]
    idle = [hip.dist_matrix(a, a) for a in inputs]
    # the idle-device result is the exact-order f32 sum (ORBmatcher.cc:1629-1650), term by term
    ref = np.zeros((n, n), np.float32)
    for k in range(128):
        d = inputs[0][:, None, k] - inputs[0][None, :, k]
        ref = ref + d * d
    np.testing.assert_array_equal(idle[0], ref)
    for k, a in enumerate(inputs):
        hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
        M = hip.dist_matrix(a, a)          # runs while ASDNet of the submission is in flight
        hip.extract_wait()
        np.testing.assert_array_equal(M, idle[k])
    hip.device_free(p)


@pytest.mark.gpu
def test_pipelined_extract_equals_sync(hip, synth):
    """asd_extract_submit / asd_extract_wait (own streams + worker thread, front half of the next frame under the
    ASDNet pass of the previous one) return exactly what asd_extract does, in submission order, also while the
    main stream is busy with other work and with the queue full."""
    frames = (5, 6, 7, 8, 9)
    imgs = [synth.scene_frame(t) for t in frames]
    ref = [hip.extract(im) for im in imgs]
    d = []
    for im in imgs:
        p = hip.device_alloc(im.nbytes)
        hip.h2d(p, im)
        d.append(p)
    # depth 1: submit / wait alternate
    for k in range(2):
        hip.extract_submit(d[k], 1241, 376, 1241, device_resident=True)
        a = synth.unit_descriptors(300, seed=k)       # unrelated work on the main stream while the extraction runs
        M = hip.dist_matrix(a, a)
        assert (np.diag(M) == 0).all()
        kps, desc = hip.extract_wait()
        np.testing.assert_array_equal(kps, ref[k][0])
        np.testing.assert_array_equal(desc, ref[k][1])
    # full queue, refilled as results are taken (the replay's steady state), frames of different content
    q = 3
    order = [0, 1, 2, 3, 4, 2, 0, 4, 1, 3, 3, 0]
    for k in order[:q]:
        hip.extract_submit(d[k], 1241, 376, 1241, device_resident=True)
    with pytest.raises(Exception):                    # a 4th outstanding submission is refused, nothing is lost
        hip.extract_submit(d[0], 1241, 376, 1241, device_resident=True)
    held = []
    for i, k in enumerate(order):
        kps, desc = hip.extract_wait(view=(i % 2 == 0))   # copies and zero-copy views alternate
        if i % 2 == 0:
            held.append((i, k, kps, desc))
        # a view stays intact while the next two submissions are made and processed
        for (j, kj, vk, vd) in held:
            if i - j <= 2:
                np.testing.assert_array_equal(vk, ref[kj][0], err_msg=f"view of submission {j} read at {i}")
                np.testing.assert_array_equal(vd, ref[kj][1], err_msg=f"view of submission {j} read at {i}")
        np.testing.assert_array_equal(kps, ref[k][0], err_msg=f"submission {i} (frame {frames[k]})")
        np.testing.assert_array_equal(desc, ref[k][1], err_msg=f"submission {i} (frame {frames[k]})")
        # the waited frame's descriptors are adoptable on the device while later frames are still in flight
        hip.frame_set(2, kps, None, (0.0, 1241.0, 0.0, 376.0))
        if i + q < len(order):
            hip.extract_submit(d[order[i + q]], 1241, 376, 1241, device_resident=True)
    with pytest.raises(Exception):                    # nothing outstanding any more
        hip.extract_wait()
    # the synchronous path still works afterwards and agrees
    kps, desc = hip.extract(imgs[1])
    np.testing.assert_array_equal(kps, ref[1][0])
    np.testing.assert_array_equal(desc, ref[1][1])
    for p in d:
        hip.device_free(p)


@pytest.mark.gpu
def test_view_lifetime_is_checkable(pkg, synth):
    """asd_extract_last_view / asd_extract_view_valid: a view stays valid -- and its contents intact -- until the submission that
    takes its buffers (ASD_EXTRACT_QUEUE + 2 = 5 submissions after its own) has been made, and the library says so."""
    ctx = pkg.AsdHip(n_features=500, max_width=640, max_height=240, max_patches=1024)
    try:
        ctx.load_weights(synth.asdnet_weights(0))
        imgs = [np.ascontiguousarray(synth.scene_frame(t)[:240, :640]) for t in range(3)]
        ref = [ctx.extract(im) for im in imgs]
        dev = []
        for im in imgs:
            p = ctx.device_alloc(im.nbytes)
            ctx.h2d(p, im)
            dev.append(p)
        imgs = dev
        assert not ctx.extract_view_valid(0)                  # nothing handed out yet
        ctx.extract_submit(imgs[0], 640, 240, 640)            # submission 0
        k0, d0 = ctx.extract_wait(view=True)
        v0 = ctx.extract_last_view()
        assert v0 == 0 and ctx.extract_view_valid(v0)
        for s in range(1, 5):                                 # submissions 1..4: the view of 0 survives all of them
            ctx.extract_submit(imgs[s % 3], 640, 240, 640)
            ctx.extract_wait(view=True)
            assert ctx.extract_last_view() == s
            assert ctx.extract_view_valid(v0), f"after submission {s}"
            np.testing.assert_array_equal(k0, ref[0][0])
            np.testing.assert_array_equal(d0, ref[0][1])
        ctx.extract_submit(imgs[1], 640, 240, 640)            # submission 5 takes the buffers of submission 0
        assert not ctx.extract_view_valid(v0)
        assert ctx.extract_view_valid(1) and ctx.extract_view_valid(4)
        assert not ctx.extract_view_valid(5)                  # not waited yet: never handed out
        ctx.extract_wait(view=True)
        assert ctx.extract_view_valid(5)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_extract_hold_delays_nothing_but_the_forward(pkg, synth):
    """asd_extract_hold: a held submission completes once the hold is lifted -- and a wait on it lifts the hold itself -- with the same results"""
    ctx = pkg.AsdHip(n_features=500, max_width=640, max_height=240, max_patches=1024)
    try:
        ctx.load_weights(synth.asdnet_weights(0))
        img = np.ascontiguousarray(synth.scene_frame(3)[:240, :640])
        ref_k, ref_d = ctx.extract(img)
        p = ctx.device_alloc(img.nbytes)
        ctx.h2d(p, img)
        ctx.extract_hold(True)                     # before any read-ahead extraction exists: nothing to hold, not an error
        ctx.extract_submit(p, 640, 240, 640)
        ctx.extract_hold(True)
        ctx.extract_submit(p, 640, 240, 640)       # its forward may be held ...
        ctx.extract_hold(False)
        for _ in range(2):
            k, d = ctx.extract_wait(view=True)
            np.testing.assert_array_equal(k, ref_k)
            np.testing.assert_array_equal(d, ref_d)
        ctx.extract_hold(True)
        ctx.extract_submit(p, 640, 240, 640)
        k, d = ctx.extract_wait(view=True)         # ... and a wait on a held submission ends the hold
        np.testing.assert_array_equal(k, ref_k)
        np.testing.assert_array_equal(d, ref_d)
        ctx.device_free(p)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_pipelined_extract_device_descriptors_adoptable(hip, oracle, synth):
    """asd_frame_set(desc=NULL) after asd_extract_wait adopts THAT frame's device descriptors even though the next
    frames' ASDNet passes are already running: checked through a matcher that reads the slot's descriptors"""
    imgs = [synth.scene_frame(t) for t in (11, 12, 13)]
    ref = [hip.extract(im) for im in imgs]
    d = []
    for im in imgs:
        p = hip.device_alloc(im.nbytes)
        hip.h2d(p, im)
        d.append(p)
    for p in d:
        hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    for k in range(3):
        kps, desc = hip.extract_wait()
        hip.frame_set(3, kps, None, bounds)            # device-resident descriptors of frame k
        hip.frame_set(4, ref[k][0], ref[k][1], bounds)  # the same frame uploaded from the host
        # SearchForInitialization-style self match: identical descriptors on both sides -> identical results
        prev = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        m_dev, n_dev, _ = hip.match_init(3, 4, prev.copy(), 30, 0.9, True)
        m_host, n_host, _ = hip.match_init(4, 4, prev.copy(), 30, 0.9, True)
        np.testing.assert_array_equal(m_dev, m_host)
        assert n_dev == n_host and n_dev > 100
    for p in d:
        hip.device_free(p)


@pytest.mark.gpu
def test_extract_500_keypoint_config(hip, oracle, synth):
    """BASELINE.json configs[0]: KITTI 00 mono at 500 keypoints / frame"""
    kps, desc = _compare_extract(hip, oracle, synth, synth.scene_frame(5), 500, override=500)
    assert 500 <= len(kps) <= 620


@pytest.mark.gpu
def test_extract_euroc_size_own_context(pkg, oracle, synth):
    """EuRoC frames (752 x 480, BASELINE.json configs[3] resolution) through a context of their own: taller than the
    shared KITTI context, 1000 features as in the reference's EuRoC launch files"""
    ctx = pkg.AsdHip(n_features=1000, max_width=752, max_height=480)
    try:
        ctx.load_weights(synth.asdnet_weights(0))
        img = synth.scene_frame(2, w=752, h=480)
        kps, desc = _compare_extract(ctx, oracle, synth, img, 1000)
        assert len(kps) >= 1000
        with pytest.raises(Exception):
            ctx.extract(synth.scene_frame(0))       # 1241 wide does not fit this context
    finally:
        ctx.close()
