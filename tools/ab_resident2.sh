set -e
mkdir -p gpurun_out/r4a
for cfg in "1 2" "1 1" "0 0" "1 2"; do
  set -- $cfg
  ASD_CHAIN_EARLY=$1 ASD_SOLVE_PRIO=$2 ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 400 --warmup 60 > gpurun_out/r4a/bench_e.json 2> gpurun_out/r4a/bench_e.err
  echo "early=$1 prio=$2"; grep -E "device clock" gpurun_out/r4a/bench_e.err | tail -1; grep -E "extract wait" gpurun_out/r4a/bench_e.err | tail -1
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_e.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'], d['roofline']['asdnet_forward_ms'])"
done
