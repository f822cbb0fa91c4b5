O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/test_asdnet.py tests/test_frontend.py -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
for cfg in "PERSIST=0" "RESERVE=0" "RESERVE=1" "RESERVE=2" "RESERVE=3" "RESERVE=4" "RESERVE=1"; do
  env ASD_ASDNET_$cfg python bench.py --cpu-frames 0 --no-lane-variant > $O/bench_$cfg.json 2>/dev/null
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', round(j['value'],1), 'fps; asdnet', round(j['roofline']['asdnet_forward_ms'],3), 'conv2', round(j['roofline']['avg_launch_us'],1))"
done
python tools/time_asdnet.py 2000 20 2>&1 | tail -3
