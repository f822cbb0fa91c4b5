"""Frame::ComputeStereoMatches (Frame.cc:360-535): oracle sanity on CPU, HIP vs oracle on the GPU."""
import numpy as np
import pytest

BOUNDS = (0.0, 1241.0, 0.0, 376.0)
MB, FX = 0.54, 718.856
MBF = MB * FX


# BASELINE configs[3]: EuRoC MH stereo, 752x480, fx = 458.654 (cameraconfig/MH_EUROC/EuRoC_config.txt), baseline 0.11 m
EUROC_W, EUROC_H, EUROC_MB, EUROC_FX = 752, 480, 0.11, 458.654


def _stereo_pair(synth, disparity=12, seed_t=0, w=1241, h=376):
    wide = synth.scene_frame(seed_t, w=w + 96, h=h)
    left = np.ascontiguousarray(wide[:, 32:32 + w])
    right = np.ascontiguousarray(wide[:, 32 + disparity:32 + disparity + w])
    return left, right


def test_oracle_stereo_recovers_constant_disparity(oracle, synth):
    left, right = _stereo_pair(synth, 12)
    exl, exr = oracle.extractor(1000), oracle.extractor(1000)
    kl, pl = exl.extract(left)
    kr, pr = exr.extract(right)
    layers = synth.asdnet_weights(0)
    # descriptors of every 1st keypoint would take the naive oracle conv minutes: use the patches' own bytes as a
    # stand-in descriptor (unit-normalised), which is all the association needs
    def fake(p):
        d = p.reshape(len(p), -1)[:, ::8].astype(np.float32)
        d -= d.mean(1, keepdims=True)
        return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    dl, dr = fake(pl), fake(pr)
    u, z, n = exl.stereo_match(exr, kl, dl, kr, dr, MB, MBF)
    ok = u >= 0
    assert n == ok.sum() and n > 0.3 * len(kl)
    disp = kl["x"][ok] - u[ok]
    assert np.abs(np.median(disp) - 12) < 0.2
    assert (np.abs(disp - 12) < 1.0).mean() > 0.95
    np.testing.assert_allclose(z[ok], np.float32(MBF) / disp, rtol=1e-6)
    assert (z[~ok] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("disparity,nfeat,w,h,mb,fx", [(12, 2000, 1241, 376, MB, FX), (40, 1000, 1241, 376, MB, FX),
                                                        (9, 2000, EUROC_W, EUROC_H, EUROC_MB, EUROC_FX)])
def test_stereo_match_parity(pkg, oracle, synth, disparity, nfeat, w, h, mb, fx):
    """the last case is BASELINE configs[3]'s size: two 752x480 contexts (left / right extractor), 2000 features each"""
    left, right = _stereo_pair(synth, disparity, seed_t=1, w=w, h=h)
    layers = synth.asdnet_weights(0)
    BOUNDS = (0.0, float(w), 0.0, float(h))
    MB, MBF = mb, mb * fx
    L = pkg.AsdHip(n_features=nfeat, max_width=w, max_height=h)
    R = pkg.AsdHip(n_features=nfeat, max_width=w, max_height=h)
    try:
        L.load_weights(layers)
        R.load_weights(layers)
        kl, dl = L.extract(left)
        kr, dr = R.extract(right)
        kl, dl, kr, dr = kl.copy(), dl.copy(), kr.copy(), dr.copy()
        L.frame_set(0, kl, dl, BOUNDS)
        L.frame_set(1, kr, dr, BOUNDS)
        gu, gz, gn = L.stereo_match(R, 0, 1, len(kl), MB, MBF)
        exl, exr = oracle.extractor(nfeat), oracle.extractor(nfeat)
        okl, _ = exl.extract(left, want_patches=False)
        okr, _ = exr.extract(right, want_patches=False)
        np.testing.assert_array_equal(okl["x"], kl["x"])          # same keypoints (front-end parity), so same inputs
        np.testing.assert_array_equal(okr["x"], kr["x"])
        eu, ez, en = exl.stereo_match(exr, kl, dl, kr, dr, MB, MBF)
        np.testing.assert_array_equal(gu, eu)
        np.testing.assert_array_equal(gz, ez)
        assert gn == en and gn > 0.3 * len(kl)
        ok = gu >= 0
        assert np.abs(np.median(kl["x"][ok] - gu[ok]) - disparity) < 0.3
    finally:
        L.close()
        R.close()


@pytest.mark.gpu
def test_stereo_match_argument_checks(pkg, synth):
    L = pkg.AsdHip(n_features=500, max_width=640, max_height=240)
    R = pkg.AsdHip(n_features=500, max_width=640, max_height=240)
    try:
        with pytest.raises(Exception):
            L.stereo_match(R, 0, 1, 0, MB, MBF)     # nothing extracted yet
    finally:
        L.close()
        R.close()
