# the local-map stage's queries by the stage-1 solver's tail (ASD_FRUSTUM_TAIL=1) against k_frustum_queries as a kernel of its own (default)
set -e
mkdir -p gpurun_out/r4a
ASD_FRUSTUM_TAIL=1 timeout -k 10 300 python -m pytest tests/test_track_chain.py tests/test_bench_host.py tests/test_matcher.py -m gpu -x -q 2>&1 | tail -1
ASD_FRUSTUM_TAIL=1 ASD_CHAIN_FUSED=0 timeout -k 10 300 python -m pytest tests/test_track_chain.py -m gpu -x -q 2>&1 | tail -1
for e in 0 1 0 1 0 1; do
  ASD_FRUSTUM_TAIL=$e ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 > gpurun_out/r4a/bench_f.json 2> gpurun_out/r4a/bench_f.err
  grep -E "device clock" gpurun_out/r4a/bench_f.err | tail -1 | cut -c1-230
  python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_f.json')); s=d['steady_state']; print('frustum_tail=$e', round(d['value'],1), round(s['ms_tracking_per_frame'],4), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],4), round(d['roofline']['asdnet_forward_ms'],4))"
done
