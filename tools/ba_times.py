"""LocalBA alone on the SURVEY 8(d) nominal problem: wall time per asd_local_ba call (ASD_TIMING=1 prints the split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(max_patches=4096)
prob = pkg.synth.ba_problem(seed=1)
for rep in range(4):
    p = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in prob.items()}
    t = time.perf_counter()
    r = hip.local_ba(p)
    print(f"rep {rep}: {1e3 * (time.perf_counter() - t):.3f} ms wall, device {hip.last_stage_ms('ba'):.3f} ms", file=sys.stderr)
