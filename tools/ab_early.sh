set -e
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_track_chain.py tests/test_bench_host.py -m gpu -x -q 2>&1 | tail -3
run() { timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(round(d['value'],1), round(s['ms_tracking_per_frame'],3), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],3))"; }
echo "early   $(run)"
echo "inorder $(ASD_CHAIN_EARLY=0 run)"
echo "early   $(run)"
echo "inorder $(ASD_CHAIN_EARLY=0 run)"
ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 400 --warmup 60 2>&1 >/dev/null | grep -E "device clock" | tail -1
