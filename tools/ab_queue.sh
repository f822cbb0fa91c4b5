set -e
mkdir -p gpurun_out/r4a
run() {
  echo "$*"
  env "$@" ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 > gpurun_out/r4a/bench_e.json 2> gpurun_out/r4a/bench_e.err || { tail -5 gpurun_out/r4a/bench_e.err; return 1; }
  grep -E "extract wait" gpurun_out/r4a/bench_e.err | tail -1
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_e.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'], d['roofline']['asdnet_forward_ms'])"
}
run ASD_BENCH_LOOKAHEAD=3
run ASD_BENCH_LOOKAHEAD=5
run ASD_BENCH_LOOKAHEAD=5 ASD_CHAIN_EARLY=2
run ASD_BENCH_LOOKAHEAD=3
run ASD_BENCH_LOOKAHEAD=5
run ASD_BENCH_LOOKAHEAD=5 ASD_CHAIN_EARLY=3
