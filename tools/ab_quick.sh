# quick check: chain/host parity tests + three bench runs (gpurun_out/r4a)
set -e
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_track_chain.py tests/test_bench_host.py tests/test_matcher.py -m gpu -x -q 2>&1 | tail -3
run() { python3 bench.py --cpu-frames 0 --no-lane-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['asdnet_forward_ms'])"; }
echo "run1 $(run)"
echo "run2 $(run)"
echo "w1   $(ASD_EXTRACT_WORKERS=1 run)"
