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


def test_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (fresh child processes, gloo rendezvous on
    127.0.0.1) and the shared sequence queue of --sequences hands every ticket out exactly once."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--selftest-dist"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["frames_total"] == 10.0
    assert sorted(k for r in j["tickets"] for k in r) == list(range(11))
    assert len(j["tickets"]) == 2 and all(len(r) > 0 for r in j["tickets"])
    # longest first: KITTI 02 (4661 frames), 00 (4541), 08 (4071) lead the queue
    assert j["queue"][:3] == ["02", "00", "08"]
    # every rank pinned itself (before any HIP call) to its own CPUs: the two sets are disjoint and non-empty
    a0, a1 = (set(a) for a in j["affinity"])
    if len(os.sched_getaffinity(0)) >= 2:
        assert a0 and a1 and not (a0 & a1), (a0, a1)


def test_world_size_mismatch_is_refused():
    """a launcher environment that disagrees with --gpus is an error, never a silent single-GPU run"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-dist"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
