"""File formats of the reference's headless replay (asd-slam_amd/host/replay_io.hpp, SURVEY 8(f) rank 3, I/O half):
camera / image config readers, KITTI image list, TUM trajectory lines, PGM frames.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "asd-slam_amd", "host", "test_replay_io")

# cameraconfig/KITTI/kitti00-02.txt of the reference (a data file: intrinsics line + Tbc line)
KITTI_00_02 = ("718.856, 718.856,  607.1928, 185.2157, 0, 0, 0, 0\n"
               "0.00007493, -0.99962055,  0.02754549, -0.00519809, -0.99994504, -0.00036369, -0.01047809,  0.06056364, "
               "0.01048413, -0.02754319, -0.99956563,  0.00379361\n")


@pytest.fixture(scope="module")
def probe():
    if not os.path.exists(PROBE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "asd-slam_amd", "csrc"), "../host/test_replay_io"])
    return lambda *a: subprocess.run([PROBE, *map(str, a)], capture_output=True, text=True)


def test_cam_info(probe, tmp_path):
    p = tmp_path / "cam.txt"
    p.write_text(KITTI_00_02)
    out = probe("cam", p).stdout.split()
    assert [float(x) for x in out[:8]] == [718.856, 718.856, 607.1928, 185.2157, 0, 0, 0, 0]
    assert out[8] == "1"
    np.testing.assert_allclose([float(x) for x in out[9:]], [0.00007493, -0.99962055, 0.02754549, -0.00519809, -0.99994504, -0.00036369,
                                                              -0.01047809, 0.06056364, 0.01048413, -0.02754319, -0.99956563, 0.00379361])
    p.write_text("707.0912, 707.09127, 601.8873, 183.1104, 0, 0, 0, 0\n")        # no camera-to-body line
    out = probe("cam", p).stdout.split()
    assert float(out[1]) == 707.09127 and out[8] == "0"
    assert probe("cam", tmp_path / "missing.txt").returncode == 1
    p.write_text("1, 2, 3\n")
    assert probe("cam", p).returncode == 1


def test_image_info(probe, tmp_path):
    p = tmp_path / "img.txt"
    p.write_text("1241,376\n2000,1.2,8\n")
    assert probe("imginfo", p).stdout.split() == ["1241", "376", "2000", "1.2", "8"]
    p.write_text("1241,376,3\n2000,1.2,8\n")
    assert probe("imginfo", p).returncode == 1


def test_kitti_image_list(probe, tmp_path):
    (tmp_path / "times.txt").write_text("0.000000e+00\n1.036224e-01\n\n2.072344e-01\n")
    lines = probe("images", tmp_path).stdout.splitlines()
    assert len(lines) == 3
    assert lines[1] == f"0.103622400 {tmp_path}/image_0/000001.png"
    assert lines[2].endswith("/image_0/000002.png")


def _eigen_quat(R):
    R = R.astype(np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    v = np.zeros(3)
    v[i] = 0.5 * s
    s = 0.5 / s
    w = (R[k, j] - R[j, k]) * s
    v[j] = (R[j, i] + R[i, j]) * s
    v[k] = (R[k, i] + R[i, k]) * s
    return np.array([v[0], v[1], v[2], w])


@pytest.mark.parametrize("rv", [(0.01, -0.02, 0.005), (2.9, 0.3, -0.2), (0.1, 3.0, 0.2), (-0.3, 0.2, 3.05)])
def test_tum_lines(probe, synth, rv):
    """System::SaveTrajectoryTUM / SaveKeyFrameTrajectoryTUM line format, incl. rotations with negative trace"""
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = synth._rot(np.array(rv)).astype(np.float32)
    T[:3, 3] = np.array([1.5, -0.25, 30.125], np.float32)
    Rwc = T[:3, :3].T.copy()
    twc = np.zeros(3, np.float32)
    for r in range(3):
        twc[r] = -np.float32(np.float32(np.float32(Rwc[r, 0] * T[0, 3]) + np.float32(Rwc[r, 1] * T[1, 3])) + np.float32(Rwc[r, 2] * T[2, 3]))
    q = _eigen_quat(Rwc).astype(np.float32)
    t = 1403636579.763555527
    args = [repr(float(x)) for x in T.reshape(-1)]
    exp = f"{t:.6f} " + " ".join(f"{float(x):.9f}" for x in list(twc) + list(q))
    assert probe("tum", repr(t), *args).stdout.strip() == exp
    exp_kf = f"{t:.6f} " + " ".join(f"{float(x):.10f}" for x in list(twc) + list(q))
    assert probe("kftum", repr(t), *args).stdout.strip() == exp_kf
    assert abs(np.linalg.norm(q) - 1) < 1e-6


def test_pgm_reader(probe, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    p = tmp_path / "f.pgm"
    p.write_bytes(b"P5\n# a comment\n53 37\n255\n" + img.tobytes())
    assert probe("pgm", p).stdout.split() == ["53", "37", str(int(img.sum()))]
    p.write_bytes(b"P5\n53 37\n255\n" + img.tobytes()[:100])       # truncated
    assert probe("pgm", p).returncode == 1
