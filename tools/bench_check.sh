set -e
O=gpurun_out/r4b; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'asdnet_ms', d['roofline']['asdnet_forward_ms'])
for k in ('steady_state','do_mapping_variant','h2d_variant','lane_variant','cpu_baseline','cpu_baseline_500'):
    v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk not in ('what','sample')} if v else None)
PY
python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20:', d['value'], d['ms_per_step'], d['steady_state']['local_ba_in_window'], d['steady_state']['frames_per_s'])"
