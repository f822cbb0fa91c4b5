"""PoseOptimization / LocalBundleAdjustment.

Pinning: tests/golden/ba_golden.npz holds outputs of the REFERENCE's own vendored g2o (compiled in place,
tests/golden/make_ba_golden.py).  CPU tests check the oracle against it; GPU tests check the HIP path
against the golden and against the oracle on larger problems.

Tolerances (fp64 state): poses |d| <= 1e-8 per quaternion/translation component, points <= 1e-6 m,
outlier / depth flags and LM iteration counts identical, chi2 relative 1e-6.  After the float32 cast the
reference applies on write-back (Converter.cc:57-71) poses agree to <= 1 ulp(f32).
"""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

POSE_ATOL = 1e-8
POINT_ATOL = 1e-6


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "ba_golden.npz"))


def ba_inputs(g, i):
    return {k: g[f"ba{i}_in_{k}"] for k in ("poses", "fixed", "points", "e_point", "e_pose", "e_obs", "e_info", "K")}


def check_ba(res, g, i):
    np.testing.assert_allclose(res["poses"], g[f"ba{i}_out_poses"], atol=POSE_ATOL, rtol=0)
    np.testing.assert_allclose(res["points"], g[f"ba{i}_out_points"], atol=POINT_ATOL, rtol=0)
    np.testing.assert_array_equal(res["edge_outlier1"], g[f"ba{i}_out_edge_outlier1"])
    np.testing.assert_array_equal(res["edge_depth_pos"], g[f"ba{i}_out_edge_depth_pos"])
    np.testing.assert_allclose(res["edge_chi2"], g[f"ba{i}_out_edge_chi2"], rtol=1e-6, atol=1e-7)
    assert res["iters_first"] == int(g[f"ba{i}_out_iters_first"])
    assert res["iters_second"] == int(g[f"ba{i}_out_iters_second"])
    np.testing.assert_allclose(res["chi2_first"], float(g[f"ba{i}_out_chi2_first"]), rtol=1e-6)
    np.testing.assert_allclose(res["chi2_second"], float(g[f"ba{i}_out_chi2_second"]), rtol=1e-6)
    # erase policy of Optimizer.cc:657-671 gives identical decisions
    bad = (res["edge_chi2"] > 5.991) | (res["edge_depth_pos"] == 0)
    gbad = (g[f"ba{i}_out_edge_chi2"] > 5.991) | (g[f"ba{i}_out_edge_depth_pos"] == 0)
    np.testing.assert_array_equal(bad, gbad)


def check_pose(pose, outlier, ninl, g, i):
    np.testing.assert_allclose(pose, g[f"pose{i}_out_pose"], atol=POSE_ATOL, rtol=0)
    np.testing.assert_array_equal(outlier, g[f"pose{i}_out_outlier"])
    assert ninl == int(g[f"pose{i}_out_ninl"])


def n_cases(g, key):
    return len(json.loads(str(g[key])))


# ------------------------------------------------------------------ oracle vs the reference's g2o (CPU)
def test_golden_inputs_match_generator(golden, synth):
    """the fixture stores inputs next to outputs; the seeded generator must still reproduce them"""
    cases = json.loads(str(golden["ba_cases"]))
    for i, kw in enumerate(cases):
        prob = synth.ba_problem(**kw)
        if kw.get("n_fixed", 1) == 0:
            prob["fixed"][0] = 1
        for k, v in ba_inputs(golden, i).items():
            np.testing.assert_array_equal(prob[k], v)


def test_oracle_local_ba_matches_g2o(oracle, golden):
    for i in range(n_cases(golden, "ba_cases")):
        check_ba(oracle.local_ba(ba_inputs(golden, i)), golden, i)


def test_oracle_pose_optimization_matches_g2o(oracle, golden):
    for i in range(n_cases(golden, "pose_cases")):
        g = golden
        pose, outlier, ninl = oracle.pose_optimize(g[f"pose{i}_in_pose"], g[f"pose{i}_in_Xw"], g[f"pose{i}_in_obs"],
                                                   g[f"pose{i}_in_info"], g[f"pose{i}_in_K"])
        check_pose(pose, outlier, ninl, g, i)


def test_oracle_vs_live_reference_nominal(oracle, synth):
    """SURVEY 8(d) nominal problem (E ~ 29k) against the reference g2o itself, when oracle/_ref is built."""
    from oracle import pyoracle
    if not pyoracle.RefG2O.available():
        pytest.skip("oracle/_ref/libg2o_ref.so not built (needs /root/reference)")
    ref = pyoracle.RefG2O()
    prob = synth.ba_problem()
    r, o = ref.local_ba(prob), oracle.local_ba(prob)
    np.testing.assert_allclose(o["poses"], r["poses"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(o["points"], r["points"], atol=1e-7, rtol=0)
    np.testing.assert_array_equal(o["edge_outlier1"], r["edge_outlier1"])
    assert (o["iters_first"], o["iters_second"]) == (r["iters_first"], r["iters_second"])


def test_pose_conversion_roundtrip(oracle, synth):
    rng = np.random.default_rng(0)
    for _ in range(50):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth._rot(rng.standard_normal(3) * 2.0).astype(np.float32)
        T[:3, 3] = rng.standard_normal(3).astype(np.float32) * 10
        p = oracle.tcw_to_pose7(T)
        assert abs(np.linalg.norm(p[:4]) - 1) < 1e-12 and p[3] >= 0
        T2 = oracle.pose7_to_tcw(p)
        np.testing.assert_allclose(T2, T, atol=3e-7)


def test_lm_reduces_cost_and_flags_outliers(oracle, synth):
    prob = synth.ba_problem(n_free=4, n_fixed=2, n_points=200, seed=3)
    res = oracle.local_ba(prob)
    assert res["chi2_second"] < res["chi2_first"] < 1e5
    # the generator's gross outliers (+-20 px) must be among the flagged edges
    assert res["edge_outlier1"].sum() >= 0.015 * len(prob["e_point"])


# ------------------------------------------------------------------ HIP (GPU)
@pytest.mark.gpu
def test_hip_local_ba_matches_g2o_golden(hip, golden):
    for i in range(n_cases(golden, "ba_cases")):
        check_ba(hip.local_ba(ba_inputs(golden, i)), golden, i)


@pytest.mark.gpu
def test_hip_pose_optimization_matches_g2o_golden(hip, golden):
    g = golden
    for i in range(n_cases(g, "pose_cases")):
        pose, outlier, ninl = hip.pose_optimize(g[f"pose{i}_in_pose"], g[f"pose{i}_in_Xw"], g[f"pose{i}_in_obs"],
                                                g[f"pose{i}_in_info"], g[f"pose{i}_in_K"])
        check_pose(pose, outlier, ninl, g, i)


@pytest.mark.gpu
def test_hip_pose_optimization_full_size(hip, oracle, synth):
    for n, seed in ((2000, 21), (4000, 22), (3, 23), (2, 24)):
        pp = synth.pose_problem(n, seed=seed, outlier_frac=0.15)
        got = hip.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        exp = oracle.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        np.testing.assert_allclose(got[0], exp[0], atol=POSE_ATOL, rtol=0)
        np.testing.assert_array_equal(got[1], exp[1])
        assert got[2] == exp[2]
        # float32 write-back (Converter::toCvMat) agrees to 1 ulp
        np.testing.assert_allclose(hip.pose7_to_tcw(got[0]), oracle.pose7_to_tcw(exp[0]), rtol=2e-7, atol=1e-7)


@pytest.mark.gpu
def test_hip_pose_optimization_is_deterministic_beside_the_extractor(hip, synth):
    """PoseOptimization runs as one workgroup that usually shares its CU with ASDNet workgroups of the read-ahead extractor.  With
    the CU's LDS queues kept full by them, a barrier that does not wait for the wave's own LDS stores lets other waves read a stale
    flag or pose: round 3 found 1-3 calls in a thousand coming back 1e-14..1e-12 off or with a garbage inlier count (hipcc had left
    the wait out in front of one barrier; asd_syncthreads() in ctx.h now writes it out, `make check-isa` checks every barrier).  The
    same problem solved 600 times beside a busy extractor must give the same bits every time (tools/diag/pose_determinism.py is the
    long form of this test)."""
    img = synth.scene_frame(0)
    d_img = hip.device_alloc(img.size)
    hip.h2d(d_img, img)
    pp = synth.pose_problem(1140, seed=5, outlier_frac=0.15)
    first, pending = None, 0
    try:
        for r in range(600):
            while pending < 2:
                hip.extract_submit(d_img, 1241, 376, 1241)
                pending += 1
            got = hip.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
            if r % 3 == 0:
                hip.extract_wait()
                pending -= 1
            if first is None:
                first = (np.array(got[0]), np.array(got[1]), got[2])
            else:
                np.testing.assert_array_equal(got[0], first[0], err_msg=f"run {r}")
                np.testing.assert_array_equal(got[1], first[1], err_msg=f"run {r}")
                assert got[2] == first[2], f"run {r}"
    finally:
        while pending:
            hip.extract_wait()
            pending -= 1


@pytest.mark.gpu
def test_hip_pose_optimization_storage_forms(hip, oracle, synth):
    """The solver keeps its edges in one of three forms (ba.hip, EdgeStore): compact LDS (observations exactly f32 and
    <= 16 distinct information values: the reference's case, and every other test here), full f64 LDS, or global memory
    (problems too large for LDS).  Inputs that are NOT f32-representable, or carry too many information values, must take
    the f64 forms and still match the oracle; sizes on both sides of the LDS limits."""
    rng = np.random.default_rng(5)
    for n, seed, kind in ((1500, 41, "obs"), (1500, 42, "info"), (2300, 43, "obs"), (4300, 44, "compact"), (4500, 45, "obs")):
        pp = synth.pose_problem(n, seed=seed, outlier_frac=0.12)
        if kind == "obs":
            pp["obs"] = pp["obs"] + rng.uniform(-1e-9, 1e-9, pp["obs"].shape)     # no longer exact in f32
            assert (pp["obs"].astype(np.float32).astype(np.float64) != pp["obs"]).any()
        elif kind == "info":
            pp["info"] = pp["info"] * (1.0 + 1e-3 * rng.integers(0, 40, pp["info"].shape))   # > 16 distinct values
            assert len(np.unique(pp["info"])) > 16
        got = hip.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        exp = oracle.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        np.testing.assert_allclose(got[0], exp[0], atol=POSE_ATOL, rtol=0)
        np.testing.assert_array_equal(got[1], exp[1])
        assert got[2] == exp[2]


@pytest.mark.gpu
def test_hip_pose_optimization_edge_cases(hip, oracle, synth):
    """fewer than 10 edges (one round only, Optimizer.cc:402), majority outliers, points behind / on the camera plane"""
    cases = []
    cases.append(synth.pose_problem(9, seed=31, outlier_frac=0.2))
    cases.append(synth.pose_problem(600, seed=32, outlier_frac=0.6))
    behind = synth.pose_problem(300, seed=33, outlier_frac=0.1)
    behind["Xw"][:20] *= -1.0                      # negative depth: large residuals, gated as outliers
    cases.append(behind)
    for pp in cases:
        got = hip.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        exp = oracle.pose_optimize(pp["pose"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
        np.testing.assert_allclose(got[0], exp[0], atol=POSE_ATOL, rtol=0)
        np.testing.assert_array_equal(got[1], exp[1])
        assert got[2] == exp[2]


@pytest.mark.gpu
def test_hip_local_ba_nominal_vs_oracle(hip, oracle, synth):
    """SURVEY 8(d) nominal LocalBA: 24 free + 12 fixed poses, 6000 points, ~29k edges."""
    prob = synth.ba_problem()
    got, exp = hip.local_ba(prob), oracle.local_ba(prob)
    np.testing.assert_allclose(got["poses"], exp["poses"], atol=POSE_ATOL, rtol=0)
    np.testing.assert_allclose(got["points"], exp["points"], atol=POINT_ATOL, rtol=0)
    np.testing.assert_array_equal(got["edge_outlier1"], exp["edge_outlier1"])
    np.testing.assert_array_equal(got["edge_depth_pos"], exp["edge_depth_pos"])
    assert (got["iters_first"], got["iters_second"]) == (exp["iters_first"], exp["iters_second"])
    np.testing.assert_allclose(got["chi2_second"], exp["chi2_second"], rtol=1e-6)
    # fixed poses are untouched bit for bit
    fx = prob["fixed"] > 0
    q = prob["poses"][fx]
    np.testing.assert_allclose(got["poses"][fx], q / np.r_[[1] * 7] , atol=1e-15)
    # determinism: fixed reduction shapes, no float atomics
    again = hip.local_ba(prob)
    for k in ("poses", "points", "edge_chi2"):
        np.testing.assert_array_equal(again[k], got[k])


@pytest.mark.gpu
def test_hip_local_ba_edge_cases(hip, oracle, synth):
    # a landmark seen only by fixed keyframes, a free keyframe that loses all its edges in round 2
    prob = synth.ba_problem(n_free=3, n_fixed=3, n_points=150, seed=12, outlier_frac=0.3)
    got, exp = hip.local_ba(prob), oracle.local_ba(prob)
    np.testing.assert_allclose(got["poses"], exp["poses"], atol=POSE_ATOL, rtol=0)
    np.testing.assert_allclose(got["points"], exp["points"], atol=POINT_ATOL, rtol=0)
    np.testing.assert_array_equal(got["edge_outlier1"], exp["edge_outlier1"])
    # invalid input is rejected, not crashed on
    bad = dict(prob)
    bad["e_pose"] = prob["e_pose"].copy(); bad["e_pose"][0] = 99
    with pytest.raises(Exception):
        hip.local_ba(bad)


@pytest.mark.gpu
def test_local_ba_structure_on_device_equals_host_structure(pkg, synth, monkeypatch):
    """The active structure of a LocalBA (h-indices, edges by landmark, pose edge lists, the Schur pair lists per block) is built by
    device kernels (k_ba_struct*); ASD_BA_STRUCT=host keeps the host loop that used to build it.  Same tables in the same order ->
    the solver forms the same sums: every output is bit-identical, on the nominal problem and on the awkward ones (landmarks seen
    only by fixed keyframes, unobserved landmarks, a pose without edges, problems with one free pose)."""
    probs = [synth.ba_problem(), synth.ba_problem(n_free=3, n_fixed=3, n_points=150, seed=12, outlier_frac=0.3),
             synth.ba_problem(n_free=1, n_fixed=2, n_points=40, seed=3, obs_per_point=2), synth.ba_problem(n_free=30, n_fixed=4, n_points=900, seed=8)]
    lonely = synth.ba_problem(n_free=4, n_fixed=2, n_points=120, seed=5)
    keep = lonely["e_pose"] != 3                      # a free pose (ids 0, 1 are the fixed ones) loses every edge; the last ten landmarks are never observed
    keep &= lonely["e_point"] < 110
    for k in ("e_pose", "e_point", "e_obs", "e_info"):
        lonely[k] = np.ascontiguousarray(lonely[k][keep])
    probs.append(lonely)
    monkeypatch.setenv("ASD_BA_STRUCT", "host")
    host = pkg.AsdHip(n_features=500, max_width=640, max_height=240, max_patches=1024)
    monkeypatch.delenv("ASD_BA_STRUCT")
    dev = pkg.AsdHip(n_features=500, max_width=640, max_height=240, max_patches=1024)
    try:
        for prob in probs:
            a, b = dev.local_ba(prob), host.local_ba(prob)
            for k in ("poses", "points", "edge_chi2", "edge_outlier1", "edge_depth_pos"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            assert (a["iters_first"], a["iters_second"], a["chi2_first"], a["chi2_second"]) == (b["iters_first"], b["iters_second"], b["chi2_first"], b["chi2_second"])
    finally:
        host.close()
        dev.close()


@pytest.mark.gpu
def test_local_ba_lane_equals_inline_and_runs_beside_tracking(hip, synth):
    """asd_local_ba_submit / _wait: LocalBundleAdjustment on the library's optional lane (the reference itself calls it in line,
    Tracking.cc:797 -> LocalMapping.cc:89).  Same kernels in the same order on another stream:
    bit-identical to asd_local_ba -- also with PoseOptimization calls of the tracking side running meanwhile -- and the
    one-run-at-a-time rules of the header are enforced."""
    prob = synth.ba_problem()
    inline = hip.local_ba(prob)
    assert hip.local_ba_poll() == 0
    with pytest.raises(Exception):
        hip.local_ba_wait()                     # nothing submitted
    pr = synth.pose_problem(n=2000, seed=0, outlier_frac=0.1)
    ref_pose = hip.pose_optimize(pr["pose"], pr["Xw"], pr["obs"], pr["info"], pr["K"])
    hip.local_ba_submit(prob)
    assert hip.local_ba_poll() in (1, 2)
    with pytest.raises(Exception):
        hip.local_ba_submit(prob)               # one run at a time
    with pytest.raises(Exception):
        hip.local_ba(prob)                      # the solver's buffers are in use
    during = [hip.pose_optimize(pr["pose"], pr["Xw"], pr["obs"], pr["info"], pr["K"]) for _ in range(8)]   # tracking side, meanwhile
    got = hip.local_ba_wait()
    assert hip.local_ba_poll() == 0
    for k in ("poses", "points", "edge_chi2", "edge_depth_pos", "edge_outlier1"):
        np.testing.assert_array_equal(got[k], inline[k])
    assert (got["iters_first"], got["iters_second"], got["chi2_second"]) == (inline["iters_first"], inline["iters_second"], inline["chi2_second"])
    for d in during:
        np.testing.assert_array_equal(d[0], ref_pose[0])
        np.testing.assert_array_equal(d[1], ref_pose[1])
    # an invalid problem is refused at submit, and the lane stays usable
    bad = dict(prob)
    bad["e_pose"] = prob["e_pose"].copy(); bad["e_pose"][0] = 99
    with pytest.raises(Exception):
        hip.local_ba_submit(bad)
    hip.local_ba_submit(prob)
    again = hip.local_ba_wait()
    np.testing.assert_array_equal(again["poses"], inline["poses"])


@pytest.mark.gpu
def test_hip_pose_conversions(hip, oracle, synth):
    rng = np.random.default_rng(5)
    for _ in range(20):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth._rot(rng.standard_normal(3) * 2.0).astype(np.float32)
        T[:3, 3] = rng.standard_normal(3).astype(np.float32) * 10
        np.testing.assert_array_equal(hip.tcw_to_pose7(T), oracle.tcw_to_pose7(T))
        p = oracle.tcw_to_pose7(T)
        np.testing.assert_array_equal(hip.pose7_to_tcw(p), oracle.pose7_to_tcw(p))
