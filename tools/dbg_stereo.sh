mkdir -p gpurun_out/r4b
echo "== tests"
timeout -k 10 500 python -m pytest tests/test_stereo.py tests/test_kitti_configs.py tests/test_frontend.py -m gpu -x -q --timeout 120 --durations=6 2>&1 | tail -16
echo "== stereo bench 300"
timeout -k 5 150 python3 bench.py --workload euroc-stereo --steps 300 --warmup 45 > gpurun_out/r4b/bench_euroc_stereo.json 2> gpurun_out/r4b/bench_euroc.err; echo "rc=$?"; tail -3 gpurun_out/r4b/bench_euroc.err; cut -c1-260 gpurun_out/r4b/bench_euroc_stereo.json
