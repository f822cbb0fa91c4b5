import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
hip = pkg.AsdHip(n_features=2000, max_width=1241, max_height=376, max_patches=4096)
hip.load_weights(pkg.synth.asdnet_weights(0))
im = pkg.synth.scene_frame(5)
ref = hip.extract(im)
p = hip.device_alloc(im.nbytes); hip.h2d(p, im)
bad = 0
for k in range(24):
    hip.extract_submit(p, 1241, 376, 1241, device_resident=True)
    a = pkg.synth.unit_descriptors(300, seed=k)
    M = hip.dist_matrix(a, a)
    if (np.diag(M) != 0).any(): bad += 1
    r = hip.extract_wait()
print('mask', hip.asdnet_split_mask(), 'bad', bad, 'of 24')
