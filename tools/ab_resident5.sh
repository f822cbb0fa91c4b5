set -e
mkdir -p gpurun_out/r4a
ASD_CHAIN_EARLY=3 timeout -k 10 300 python -m pytest tests/test_track_chain.py tests/test_bench_host.py -m gpu -x -q 2>&1 | tail -2
run() {
  echo "$*"
  env "$@" ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 > gpurun_out/r4a/bench_e.json 2> gpurun_out/r4a/bench_e.err
  grep -E "device clock" gpurun_out/r4a/bench_e.err | tail -1; grep -E "extract wait|shares a hardware" gpurun_out/r4a/bench_e.err | tail -2
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_e.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'], d['roofline']['asdnet_forward_ms'])"
}
run ASD_CHAIN_EARLY=3
run ASD_CHAIN_EARLY=0
run ASD_CHAIN_EARLY=3
run ASD_CHAIN_EARLY=0
