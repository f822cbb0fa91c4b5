"""The library must not contain packed-f32 vector arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).

Measured on MI355X with a stand-alone program (tools/ubench/mfma_pk_hazard.hip, profiles/r02_mfma_pk_hazard.txt): a
v_pk_add_f32 whose src1 carries op_sel:[0,1] -- the form the SLP vectoriser emits when both halves use the same scalar --
loses its low-half result in lanes 48-63 while a wave of ANY kernel that issues bf16 MFMAs is resident on the same CU.  The
ASDNet kernels run concurrently with every other kernel of the library in the read-ahead pipeline, so the build disables
the SLP vectoriser; this test compiles every kernel source to device assembly with the build's flags (`make check-isa`, a
cross-compile: no GPU needed) and fails if a packed-f32 arithmetic instruction appears.

The same target runs tools/check_barriers.py: every s_barrier of the device code must have `s_waitcnt lgkmcnt(0)` in front of it
in its own basic block.  hipcc (ROCm 7.2) left that wait out in front of one barrier of k_pose_opt; beside the extractor's ASDNet
workgroups 1-3 PoseOptimization calls in a thousand then returned a slightly different pose or a garbage inlier count (ctx.h,
asd_syncthreads; tests/test_optimizer.py::test_hip_pose_optimization_is_deterministic_beside_the_extractor)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_packed_f32_instructions_in_any_kernel():
    csrc = os.path.join(ROOT, "asd-slam_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j4", "check-isa"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no packed-f32 instructions" in r.stdout
    assert "0 without the LDS wait" in r.stdout
