import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(max_patches=4096); hip.load_weights(pkg.synth.asdnet_weights(0))
hip.describe(pkg.synth.random_patches(2000, seed=5))
for mode, name in enumerate(["full", "no act staging", "no MFMA", "no epilogue stores"]):
    ms = C.c_float()
    rc = hip.lib.asd_debug_conv2_ablate(hip.ctx, 2000, mode, 20, C.byref(ms)); assert rc == 0
    print(f"conv2 {name:20s} {ms.value*1e3:8.1f} us")
