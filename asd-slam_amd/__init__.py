"""asd-slam_amd: MI355X-native implementation of ASD-SLAM's per-frame hot path.

The product is `libasdhip.so` (C ABI, include/asd_slam.h).  This package only holds the
HIP/C++ sources (csrc/), the C++ host-side mirror of the reference call surface (host/),
a thin ctypes binding used by tests / bench (capi.py) and seeded synthetic inputs
(synth.py).  The directory name has a hyphen, so load it with `load_package()` below or
via importlib; tests/conftest.py and bench.py do that.
"""
from . import capi, synth  # noqa: F401
from .capi import AsdHip, AsdError, lib_path  # noqa: F401
