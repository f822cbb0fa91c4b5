# the round-end checks in one gpurun call: GPU tests, smoke, default bench (gpurun_out/r5a)
set -e
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5a/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'asdnet_ms', d['roofline']['asdnet_forward_ms'])
for k in ('steady_state','one_submission_variant','do_mapping_variant','h2d_variant','lane_variant','cpu_baseline','cpu_baseline_500'):
    v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk not in ('what','sample')} if v else None)
PY
