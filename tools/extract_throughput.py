"""Throughput of the read-ahead extractor alone (no tracking): frames/s with the submission queue kept full."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(pkg.synth.asdnet_weights(0))
frames = []
for t in range(8):
    f = pkg.synth.scene_frame(t)
    p = hip.device_alloc(f.nbytes); hip.h2d(p, f); frames.append(p)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def run(n):
    q = 0
    for i in range(min(depth, n)):
        hip.extract_submit(frames[i % 8], 1241, 376, 1241); q += 1
    t0 = time.perf_counter()
    for i in range(n):
        hip.extract_wait(view=True)
        if i + depth < n:
            hip.extract_submit(frames[(i + depth) % 8], 1241, 376, 1241)
    return time.perf_counter() - t0
run(20)
dt = run(100)
print(f"queue depth {depth}: {100 / dt:.1f} frames/s, {1e3 * dt / 100:.3f} ms/frame (sequential asd_extract: ", end="")
t0 = time.perf_counter()
for i in range(30): hip.extract_device(frames[i % 8], 1241, 376, 1241)
print(f"{1e3 * (time.perf_counter() - t0) / 30:.3f} ms/frame)")
