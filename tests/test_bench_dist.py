"""N>1 control plane of bench.py rehearsed with 2 gloo ranks on the CPU (no GPU work)."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT


def test_two_rank_gloo_rendezvous():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7",
           "--selftest-dist"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["frames_total"] == 14.0
    assert j["t_max"] >= 0.1  # the slower rank (0.05 * 2) bounds the job
