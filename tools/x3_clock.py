"""In-kernel clock and MFMA-loop cycles of the split-operand ASDNet layers (diagnostic stamps, see asd_debug_x3_clock)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hip = pkg.AsdHip(max_patches=max(4096, n))
hip.load_weights(pkg.synth.asdnet_weights(0))
patches = pkg.synth.random_patches(n, seed=5)
dp = hip.device_alloc(n * 1024); dd = hip.device_alloc(n * 512)
hip.h2d(dp, patches)
hip.describe_timed(dp, n, dd, 20)          # real activations in both buffers, chip warm
# the debug entry reads ctx->d_patches: describe once through the host path to fill it
hip.describe(patches)
f = hip.lib.asd_debug_x3_clock
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
for layer in (2, 3, 4, 5, 6):
    cyc, ghz = C.c_double(), C.c_double()
    rc = f(hip.ctx, layer, n, 200, C.byref(cyc), C.byref(ghz)); assert rc == 0, rc
    print(f"conv{layer}: MFMA loop {cyc.value:.0f} shader cycles per workgroup, in-kernel clock {ghz.value:.2f} GHz")
