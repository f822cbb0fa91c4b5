"""Device-time of the ASDNet forward at N patches (resident inputs, hipEvents on the ctx stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hip = pkg.AsdHip(max_patches=max(4096, n))
hip.load_weights(pkg.synth.asdnet_weights(0))
patches = pkg.synth.random_patches(n, seed=5)
dp = hip.device_alloc(n * 1024); dd = hip.device_alloc(n * 512)
hip.h2d(dp, patches)
hip.describe_timed(dp, n, dd, 3)
ms = hip.describe_timed(dp, n, dd, reps)
tf = n * pkg.synth.ASDNET_FLOP_PER_PATCH / (ms * 1e-3) / 1e12
hip.profile_enable(True); hip.describe_timed(dp, n, dd, 10)
print("  per-layer us:", " ".join(f"{1e3*hip.profile_get(l)[0]/max(hip.profile_get(l)[1],1):.0f}" for l in range(8)))
print(f"asdnet N={n}: {ms:.3f} ms/forward, {tf:.1f} TFLOP/s f32 ({tf/157.3*100:.1f}% of 157.3 TF peak)")
