set -e
O=gpurun_out/r4a; mkdir -p $O
run() { python3 bench.py --cpu-frames 0 --no-lane-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['asdnet_forward_ms'])"; }
echo "default  $(run)"
echo "maxwg2   $(ASD_X3_MAXWG=2 run)"
echo "default  $(run)"
echo "maxwg2   $(ASD_X3_MAXWG=2 run)"
echo "maxwg1   $(ASD_X3_MAXWG=1 run)"
