set -e
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_frontend.py tests/test_kitti_configs.py tests/test_bench_host.py -m gpu -x -q 2>&1 | tail -3
run() { python3 bench.py --cpu-frames 0 --no-lane-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "workers2 $(run)"
echo "workers1 $(ASD_EXTRACT_WORKERS=1 run)"
echo "workers2 $(run)"
echo "workers2 la3 $(ASD_BENCH_LOOKAHEAD=3 run)"
ASD_TIMING=1 python3 bench.py --cpu-frames 0 --no-lane-variant --steps 300 --warmup 60 2>&1 >/dev/null | grep -E "extract worker|track_loop\] steps" | tail -4
