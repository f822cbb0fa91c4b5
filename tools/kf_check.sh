set -e
mkdir -p gpurun_out/r4b
timeout -k 10 500 python -m pytest tests/test_mapping.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 700 python3 tools/kf_times.py --reps 12 > gpurun_out/r4b/kf_ops.json 2> gpurun_out/r4b/kf_ops.err || { tail -5 gpurun_out/r4b/kf_ops.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r4b/kf_ops.json'))['kf_ops']
for k in ('create_new_map_points_batch','search_in_neighbors_fuse_batch','distinctive_descriptor_batch','per_keyframe_stage_batched_ms'): print(k, d[k])"
python3 bench.py --cpu-frames 0 > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err || { tail -5 gpurun_out/r4b/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k in ('steady_state','do_mapping_variant','h2d_variant','lane_variant'):
    v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk not in ('what','sample')} if v else None)
PY
