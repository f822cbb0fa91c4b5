set -e
timeout -k 10 600 python -m pytest tests/test_frontend.py tests/test_kitti_configs.py tests/test_stereo.py tests/test_replay_tool.py -m gpu -x -q 2>&1 | tail -3
bash tools/bench_check.sh
