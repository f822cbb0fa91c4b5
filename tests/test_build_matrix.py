"""Run-time switches that select a fallback path are part of the tested matrix: a subset of the GPU parity tests is repeated in
a child process under each switch (the switches are read once per process)."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

CASES = [
    ({"ASD_RESULT_COPY": "1", "ASD_UPLOAD_COPY": "1"}, ["tests/test_track_chain.py", "tests/test_matcher.py::test_host_and_device_replay_agree"]),
    ({"ASD_UPLOAD_SEPARATE": "1"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_FRONT_PRIO_MID": "1"}, ["tests/test_bench_host.py"]),
    # asd_track_frame in its resident form (one solver kernel per frame launched ahead, tickets between the streams; opt-in), replay and
    # solver as separate kernels, the old bid-based claim replay, one extraction worker
    ({"ASD_CHAIN_EARLY": "1"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_CHAIN_EARLY": "2"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_CHAIN_EARLY": "3"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    # ... and with kernels launched ahead that give up after ~20 us: every frame finds its kernel gone and launches a fresh one
    ({"ASD_CHAIN_EARLY": "1", "ASD_SOLVER_IDLE_POLLS": "20"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_CHAIN_EARLY": "2", "ASD_SOLVER_IDLE_POLLS": "20"}, ["tests/test_bench_host.py"]),
    ({"ASD_CHAIN_FUSED": "0"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_FRUSTUM_TAIL": "1"}, ["tests/test_track_chain.py", "tests/test_bench_host.py"]),
    ({"ASD_RESOLVE": "bids"}, ["tests/test_matcher.py", "tests/test_track_chain.py::test_track_motion_model_equals_separate_calls"]),
    ({"ASD_EXTRACT_WORKERS": "1"}, ["tests/test_bench_host.py", "tests/test_kitti_configs.py"]),
    ({"ASD_ASDNET_PERSIST": "1", "ASD_ASDNET_RESERVE": "1"}, ["tests/test_asdnet.py", "tests/test_frontend.py::test_extract_kitti_size_bit_exact"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("env,targets", CASES, ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else None)
def test_switch(env, targets):
    if os.environ.get("ASD_BUILD_MATRIX_CHILD"):
        pytest.skip("already inside a build-matrix child")
    e = dict(os.environ, ASD_BUILD_MATRIX_CHILD="1", **env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *targets],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
