"""Time conv2 (fused) / conv4 / conv6 with phases removed (bits: 1 no staging, 2 no MFMA, 4 no epilogue stores)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(max_patches=4096); hip.load_weights(pkg.synth.asdnet_weights(0))
hip.describe(pkg.synth.random_patches(2000, seed=5))
names = {13: "loop -barrier", 21: "loop -wstream", 29: "loop -both", 0: "full", 1: "no staging", 4: "no epilogue", 5: "MFMA loop only", 2: "no MFMA", 6: "staging only", 3: "epilogue only", 7: "nothing"}
for layer in (2, 4, 6):
    row = []
    for mode in (0, 1, 4, 5, 13, 21, 29, 2, 6, 3, 7):
        ms = C.c_float()
        rc = hip.lib.asd_debug_conv_ablate(hip.ctx, layer, 2000, mode, 10, C.byref(ms)); assert rc == 0
        row.append(f"{names[mode]}={ms.value*1e3:.0f}")
    print(f"conv{layer}: " + "  ".join(row))
