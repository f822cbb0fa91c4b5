set -e
run() { python3 bench.py --cpu-frames 0 --no-lane-variant --steps 450 --warmup 60 "$@" 2>> gpurun_out/r4a/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
mkdir -p gpurun_out/r4a
echo "la2 $(run)"
echo "la3 $(ASD_BENCH_LOOKAHEAD=3 run)"
echo "la2 $(run)"
echo "la3 $(ASD_BENCH_LOOKAHEAD=3 run)"
