set -e
O=gpurun_out/r4a; mkdir -p $O
run() { timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(round(d['value'],1), round(s['ms_tracking_per_frame'],3), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],3), round(d['roofline']['asdnet_forward_ms'],3))"; }
echo "early la2      $(run)"
echo "early la3      $(ASD_BENCH_LOOKAHEAD=3 run)"
echo "early la3 w1   $(ASD_BENCH_LOOKAHEAD=3 ASD_EXTRACT_WORKERS=1 run)"
echo "early la2 w1   $(ASD_EXTRACT_WORKERS=1 run)"
echo "early la3      $(ASD_BENCH_LOOKAHEAD=3 run)"
