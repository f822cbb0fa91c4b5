"""Library-level view of the round-1 co-residency failure: asd_dist_matrix (caller's stream) while the read-ahead
extractor runs ASDNet on its own streams.  Run once per library build:

    ASDHIP_LIB=asd-slam_amd/libasdhip_slp.so python tools/diag_coresidency.py     # SLP-vectorised victims, 16x16x32 MFMA
    ASDHIP_LIB=asd-slam_amd/libasdhip_slp32.so python tools/diag_coresidency.py   # SLP-vectorised victims, 32x32x16 MFMA
    python tools/diag_coresidency.py                                              # shipped build

Prints, per build, how many of the concurrent asd_dist_matrix calls returned a value that differs from the same call made on
an idle device, and where the wrong values sit (row, column, lane of the 256-thread workgroup, bits that differ).
The stand-alone form of the same question, without any library code, is tools/ubench/mfma_pk_hazard.hip.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(pkg.synth.asdnet_weights(0))
im = pkg.synth.scene_frame(5)
hip.extract(im)
p = hip.device_alloc(im.nbytes)
hip.h2d(p, im)
N = 300
ROUNDS = int(os.environ.get("DIAG_ROUNDS", "24"))
clean = []
for k in range(ROUNDS):
    a = pkg.synth.unit_descriptors(N, seed=k)
    clean.append(hip.dist_matrix(a, a))
# the idle-device result must itself be right: exact-order f32 sum on the host
a0 = pkg.synth.unit_descriptors(N, seed=0)
ref = np.zeros((N, N), np.float32)
for k in range(128):
    d = (a0[:, None, k] - a0[None, :, k]).astype(np.float32)
    ref = (ref + d * d).astype(np.float32)
idle_ok = bool(np.array_equal(ref, clean[0]))
bad_calls, details = 0, []
for k in range(ROUNDS):
    hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
    a = pkg.synth.unit_descriptors(N, seed=k)
    M = hip.dist_matrix(a, a)
    hip.extract_wait()
    w = np.argwhere(M != clean[k])
    if len(w):
        bad_calls += 1
        if len(details) < 6:
            cols = sorted(set(int(c) for c in w[:, 1]))
            rows = sorted(set(int(r) for r in w[:, 0]))
            ex = [(int(r), int(c), float(M[r, c]), float(clean[k][r, c]),
                   hex(int(M[r, c].view(np.uint32) ^ clean[k][r, c].view(np.uint32)))) for r, c in w[:8]]
            details.append({"call": k, "wrong": int(len(w)), "rows": rows[:40], "cols": cols[:80],
                            "lanes_mod64": sorted(set(c % 64 for c in cols)), "examples": ex})
out = {"lib": os.path.basename(pkg.lib_path()), "split_mask": hip.asdnet_split_mask(), "idle_result_equals_host_sum": idle_ok,
       "concurrent_calls": ROUNDS, "calls_with_wrong_values": bad_calls, "details": details}
print(json.dumps(out))
hip.close()
