"""host/asd_replay: headless replay of an image sequence (front-end half; Examples/Monocular/kitti.cc:116-155
with the tracker reduced to ExtractDesc + grid + frame-to-frame matchers).  Argument handling on the CPU; on the GPU the
per-frame keypoint and match counts of a six-frame synthetic sequence against the same calls made through the binding."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

TOOL = os.path.join(ROOT, "asd-slam_amd", "host", "asd_replay")
K = (718.856, 718.856, 607.1928, 185.2157)


def _write_sequence(tmp, synth, n):
    seq = tmp / "seq"
    (seq / "image_0").mkdir(parents=True)
    frames = [synth.scene_frame(40 + t) for t in range(n)]
    with open(seq / "times.txt", "w") as f:
        for t in range(n):
            f.write(f"{0.1 * t:.6e}\n")
    for t, im in enumerate(frames):
        with open(seq / "image_0" / f"{t:06d}.pgm", "wb") as f:
            f.write(b"P5\n# synthetic\n%d %d\n255\n" % (im.shape[1], im.shape[0]) + im.tobytes())
    cam = tmp / "cam.txt"
    cam.write_text("%r, %r,  %r, %r, 0, 0, 0, 0\n" % K)
    layers = synth.asdnet_weights(0)
    with open(tmp / "weights.bin", "wb") as f:
        for w, m, v in layers:
            f.write(np.ascontiguousarray(w, np.float32).tobytes() + np.ascontiguousarray(m, np.float32).tobytes() +
                    np.ascontiguousarray(v, np.float32).tobytes())
    return str(seq), str(cam), str(tmp / "weights.bin"), frames


def test_replay_tool_usage_and_input_errors(tmp_path):
    assert os.path.exists(TOOL), "asd_replay not built: run __graft_entry__.build()"
    p = subprocess.run([TOOL], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
    p = subprocess.run([TOOL, str(tmp_path / "nope"), "cam.txt", "w.bin"], capture_output=True, text=True)
    assert p.returncode == 2 and "times.txt" in p.stderr


@pytest.mark.gpu
def test_replay_tool_matches_the_binding(tmp_path, pkg, synth):
    n = 6
    seq, cam, weights, frames = _write_sequence(tmp_path, synth, n)
    stats = str(tmp_path / "stats.csv")
    p = subprocess.run([TOOL, seq, cam, weights, "--stats", stats, "--tum", str(tmp_path / "traj.txt")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [l.strip().split(",") for l in open(stats).read().splitlines()[1:]]
    assert len(rows) == n
    hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4000)
    try:
        hip.load_weights(synth.asdnet_weights(0))
        last = None
        for t, im in enumerate(frames):
            kps, desc = hip.extract(im)
            kps, desc = kps.copy(), desc.copy()
            hip.frame_set(t & 1, kps, desc, (0.0, 1241.0, 0.0, 376.0))
            nm = 0
            if last is not None:
                pm = np.stack([last["x"], last["y"]], 1).astype(np.float32)
                _, nm, _ = hip.match_init((t & 1) ^ 1, t & 1, pm, 100, 0.9, True)
            assert int(rows[t][2]) == len(kps) and int(rows[t][3]) == nm, (t, rows[t], len(kps), nm)
            last = kps
    finally:
        hip.close()
    # the same sequence with identity poses: SearchByProjection(cur, last) through the descriptor bank
    poses = tmp_path / "poses.txt"
    poses.write_text("1 0 0 0 0 1 0 0 0 0 1 0\n" * n)
    p = subprocess.run([TOOL, seq, cam, weights, "--stats", stats, "--poses", str(poses), "--lookahead", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows2 = [l.strip().split(",") for l in open(stats).read().splitlines()[1:]]
    assert [r[2] for r in rows2] == [r[2] for r in rows]           # keypoints do not depend on the read-ahead depth
    assert all(int(r[3]) > 200 for r in rows2[1:])                  # a 3 px drift at 20 m is inside the 15 px windows
    traj = open(tmp_path / "traj.txt").read().splitlines()
    # identity pose: twc = -(Rwc * tcw) prints as -0.000000000 with the reference's `fixed` stream too (System.cc:529)
    assert len(traj) == n and [abs(float(v)) for v in traj[0].split()[1:]] == [0.0] * 6 + [1.0]
    assert all(len(v.split(".")[1]) == 9 for v in traj[0].split()[1:]) and len(traj[0].split()[0].split(".")[1]) == 6


@pytest.mark.gpu
def test_replay_chain_carries_the_metric(tmp_path, synth):
    """asd_replay --chain: the metric's per-frame chain (extract, grid, both tracking stages, in-line LocalBA every --max_step_KF frames)
    over an image sequence read from disk, reported as one JSON line with bench.py's keys."""
    import json
    n = 24
    seq, cam, weights, _ = _write_sequence(tmp_path, synth, n)
    p = subprocess.run([TOOL, seq, cam, weights, "--chain", "--max_step_KF", "5", "--warmup", "4"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"].startswith("frames/sec end-to-end tracking+LocalBA") and j["unit"] == "frames/s"
    assert j["steps"] == n - 4 and j["warmup"] == 4 and j["value"] > 10 and abs(j["value"] * j["ms_per_step"] - 1000.0) < 1.0
    assert j["config"]["local_ba"].startswith("in line (reference order)") and j["config"]["kf_interval"] == 5
    assert j["roofline"]["achieved"] > 0 and 0 < j["roofline"]["frac"] < 1
    ls = j["last_step"]
    assert ls["n_kp"] >= 2000 and ls["m1"] > 300 and ls["inliers"] > 300    # a 3 px drift is inside the 15 px windows of the identity prediction
