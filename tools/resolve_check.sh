# matcher / chain tests, then the resolve stamps and the headline of one bench run
mkdir -p gpurun_out/r4a
timeout -k 10 200 python -m pytest tests/test_matcher.py tests/test_track_chain.py -m gpu -x -q 2>&1 | tail -1
ASD_TIMING=1 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 400 --warmup 60 > gpurun_out/r4a/bench_t.json 2> gpurun_out/r4a/bench_t.err
grep -E "track_frame resolve kind|device clock" gpurun_out/r4a/bench_t.err | tail -3
python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench_t.json')); print(d['value'], d['steady_state']['ms_tracking_per_frame'], d['steady_state']['ms_per_local_ba'])"
