"""Run-time switches that select a fallback path are part of the tested matrix: a subset of the GPU parity tests is repeated in
a child process under each switch (the switches are read once per process)."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

CASES = [
    # copy commands instead of copy kernels for the tracking stages' upload blocks
    ({"ASD_UPLOAD_COPY": "1"}, ["tests/test_track_chain.py", "tests/test_matcher.py::test_host_and_device_replay_agree"]),
    # one extraction worker instead of two
    ({"ASD_EXTRACT_WORKERS": "1"}, ["tests/test_bench_host.py", "tests/test_kitti_configs.py"]),
    # the experimental LDS-image / weight-ring ASDNet kernels (asdnet_ring.hip) through the whole extractor
    ({"ASD_ASDNET_RING": "6"}, ["tests/test_asdnet.py", "tests/test_frontend.py::test_extract_kitti_size_bit_exact"]),
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
