"""The C ABI: libasdhip.so loads without a GPU and exports every function include/asd_slam.h declares;
the product never links the oracle."""
import ctypes
import os
import re
import subprocess

from tests.conftest import ROOT, load_package


def declared_functions():
    src = open(os.path.join(ROOT, "include", "asd_slam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(asd_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ("asd_ctx_create", "asd_load_weights", "asd_describe", "asd_extract", "asd_frame_set",
                 "asd_match_project_frame", "asd_match_project_points", "asd_match_init", "asd_dist_matrix",
                 "asd_pose_optimize", "asd_local_ba", "asd_frustum", "asd_distinctive_descriptor"):
        assert must in names


def test_library_exports_every_declared_symbol():
    pkg = load_package()
    path = pkg.lib_path()
    assert os.path.exists(path), "libasdhip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(path)  # loads without a GPU; no compute call is made here
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/asd_slam.h but not exported: {missing}"
    lib.asd_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.asd_version()


def test_product_does_not_link_the_oracle():
    pkg = load_package()
    out = subprocess.run(["ldd", pkg.lib_path()], capture_output=True, text=True).stdout
    assert "liboracle" not in out and "g2o_ref" not in out
    syms = subprocess.run(["nm", "-D", pkg.lib_path()], capture_output=True, text=True).stdout
    assert " orc_" not in syms and "ref_local_ba" not in syms
    # and no product source mentions the oracle directory
    csrc = os.path.join(ROOT, "asd-slam_amd")
    for dp, _dn, fns in os.walk(csrc):
        for fn in fns:
            if fn.endswith((".hip", ".cpp", ".h", ".py")):
                txt = open(os.path.join(dp, fn)).read()
                assert "liboracle" not in txt and "pyoracle" not in txt, fn


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device ctx creation must raise (on the GPU box it succeeds)."""
    pkg = load_package()
    try:
        ctx = pkg.AsdHip(n_features=500, max_width=320, max_height=240, max_patches=1000)
    except pkg.AsdError as e:
        assert e.code == -2
    else:
        ctx.close()
