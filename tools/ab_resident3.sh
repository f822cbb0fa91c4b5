set -e
mkdir -p gpurun_out/r4a
run() {
  echo "$*"
  env "$@" ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 400 --warmup 60 > gpurun_out/r4a/bench_e.json 2> gpurun_out/r4a/bench_e.err
  grep -E "device clock" gpurun_out/r4a/bench_e.err | tail -1; grep -E "extract wait|shares a hardware" gpurun_out/r4a/bench_e.err | tail -2
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_e.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'], d['roofline']['asdnet_forward_ms'])"
}
run ASD_CHAIN_EARLY=1 ASD_SOLVE_PRIO=2 ASD_FRONT_PRIO_LOW=1
run ASD_CHAIN_EARLY=0 ASD_FRONT_PRIO_LOW=1
run ASD_CHAIN_EARLY=1 ASD_SOLVE_PRIO=2 ASD_FRONT_PRIO_LOW=1 GPU_MAX_HW_QUEUES=8
run ASD_CHAIN_EARLY=1 ASD_SOLVE_PRIO=0 GPU_MAX_HW_QUEUES=8
