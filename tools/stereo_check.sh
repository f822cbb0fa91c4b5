set -e
mkdir -p gpurun_out/r4b
timeout -k 10 600 python -m pytest tests/test_bench_host.py tests/test_stereo.py tests/test_kitti_configs.py -m gpu -x -q 2>&1 | tail -15
python3 bench.py --workload euroc-stereo --steps 300 --warmup 45 > gpurun_out/r4b/bench_euroc_stereo.json 2> gpurun_out/r4b/bench_euroc.err || { tail -5 gpurun_out/r4b/bench_euroc.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r4b/bench_euroc_stereo.json').read().strip().splitlines()[-1]); print('stereo cxx', d['value'], d['ms_per_step'], d['last_step'])"
python3 bench.py --workload euroc-stereo --host python --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stereo python', d['value'], d['ms_per_step'])"
