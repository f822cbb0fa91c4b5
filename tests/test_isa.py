"""The library must not contain packed-f32 vector arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).

Measured on MI355X (asdnet.hip, ASD_X3_S16): a wave executing those instructions returns wrong values in groups of 16
lanes while another wave on the same CU issues v_mfma_f32_16x16x32_bf16 -- the MFMA shape of the ASDNet kernels, which run
concurrently with every other kernel of the library in the read-ahead pipeline.  The build therefore disables the SLP
vectoriser; this test compiles every kernel source to device assembly with the build's flags (`make check-isa`, a
cross-compile: no GPU needed) and fails if such an instruction appears."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_packed_f32_instructions_in_any_kernel():
    csrc = os.path.join(ROOT, "asd-slam_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j4", "check-isa"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no packed-f32 instructions" in r.stdout
