import sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("asd_bench", "bench.py"); bench = importlib.util.module_from_spec(spec); sys.modules["asd_bench"]=bench; spec.loader.exec_module(bench)
import __graft_entry__ as g
pkg = g.load_package()
def run(fused, n=20):
    wl = bench.Workload(pkg.synth)
    be = bench.HipBackend(pkg, wl, 0, pipeline=False)
    be.fused = fused
    out=[]; last=None
    for t in range(n):
        last, st = bench.run_steps_python(be, wl, t, 1, last, prefetch_beyond=False)
        out.append({k:v for k,v in st.items() if k!='ba_chi2'})
    be.close()
    return out
a=run(True); b=run(False)
for t,(x,y) in enumerate(zip(a,b)):
    if x!=y: print("mono frame",t,x,y)
print("mono fused == nonfused:", a==b)
