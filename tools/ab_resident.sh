# asd_track_frame: the resident (ticket hand-over) form against the in-order fused form -- chain tests under both, then the bench
set -e
mkdir -p gpurun_out/r4a
ASD_CHAIN_EARLY=1 timeout -k 10 300 python -m pytest tests/test_track_chain.py tests/test_bench_host.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_track_chain.py tests/test_bench_host.py -m gpu -x -q 2>&1 | tail -1
for e in 1 0 1 0; do
  ASD_CHAIN_EARLY=$e ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 400 --warmup 60 > gpurun_out/r4a/bench_e$e.json 2> gpurun_out/r4a/bench_e$e.err
  echo "early=$e"; grep -E "device clock" gpurun_out/r4a/bench_e$e.err | tail -1
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_e$e.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'], d['roofline']['asdnet_forward_ms'])"
done
