"""Wall-clock per API call of the bench step (host view), 30 frames."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_pipe = '--pipeline' in sys.argv
sys.argv = ['bench.py'] + (['--pipeline'] if _pipe else [])
import bench, numpy as np
pkg = bench.graft.load_package()
wl = bench.Workload(pkg.synth)
import sys as _s
be = bench.HipBackend(pkg, wl, 0, pipeline=('--pipeline' in _s.argv))
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(name):
    f = getattr(be, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
    setattr(be, name, g)
for n in ('extract', 'prefetch', 'set_map_descriptors', 'make_frame', 'match_frame', 'pose_opt', 'frustum', 'match_points', 'local_ba'):
    wrap(n)
last, _ = bench.run_steps(be, wl, 0, 15, None, prefetch_beyond=True)
acc.clear(); cnt.clear()
t0 = time.perf_counter()
last, st = bench.run_steps(be, wl, 15, 30, last, prefetch_beyond=True)
tot = time.perf_counter() - t0
print('total ms/frame', 1e3 * tot / 30, 'host cores', bench.host_cores(), 'cpu_count', os.cpu_count())
for k in acc: print(f'{k:14s} {1e3*acc[k]/30:8.3f} ms/frame  ({cnt[k]} calls, {1e3*acc[k]/cnt[k]:.3f} ms/call)')
print('python glue     %.3f ms/frame' % (1e3 * (tot - sum(acc.values())) / 30))
print('device ms: extract', be.hip.last_stage_ms('extract'), 'asdnet', be.hip.last_stage_ms('asdnet'), 'ba', be.hip.last_stage_ms('ba'))
