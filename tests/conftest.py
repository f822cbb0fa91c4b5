import importlib.util
import os
import faulthandler
import sys

if os.environ.get("ASD_TEST_WATCHDOG"):   # diagnostics: Python stacks of all threads after N seconds (a call stuck in C code never returns to pytest-timeout)
    _wd = open(os.environ.get("ASD_TEST_WATCHDOG_FILE", "/tmp/asd_watchdog.txt"), "w")
    faulthandler.dump_traceback_later(float(os.environ["ASD_TEST_WATCHDOG"]), exit=True, file=_wd)
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The package directory is `asd-slam_amd` (hyphen), so import it by path."""
    if "asd_slam_amd" in sys.modules:
        return sys.modules["asd_slam_amd"]
    pkg_dir = os.path.join(ROOT, "asd-slam_amd")
    spec = importlib.util.spec_from_file_location("asd_slam_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["asd_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    po = load_oracle()
    po.build()
    return po.Oracle()


@pytest.fixture(scope="session")
def oracle_mod():
    """the oracle's Python module (RefG2O / RefDBoW2: the reference's own sources compiled in place, where present)"""
    return load_oracle()


@pytest.fixture(scope="session")
def hip(pkg):
    """One HIP context shared by the GPU tests (fails loudly if libasdhip / the GPU is missing)."""
    ctx = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
    ctx.load_weights(pkg.synth.asdnet_weights(0))
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def asdnet_golden():
    return np.load(os.path.join(GOLDEN, "asdnet_golden.npz"))
