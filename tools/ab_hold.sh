# LocalBA in line with the extractor standing back (asd_extract_hold, default) against the extractor running on: alternating on one box
set -e
mkdir -p gpurun_out/r4a
timeout -k 10 300 python -m pytest tests/test_bench_host.py tests/test_frontend.py -m gpu -x -q 2>&1 | tail -1
for e in 1 0 1 0 1 0; do
  ASD_BA_HOLD_EXTRACT=$e ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 > gpurun_out/r4a/bench_f.json 2> gpurun_out/r4a/bench_f.err
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_f.json')); s=d['steady_state']; print('hold=$e', round(d['value'],1), round(s['ms_tracking_per_frame'],4), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],4), round(d['roofline']['asdnet_forward_ms'],4))"
done
