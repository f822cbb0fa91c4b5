# Host gaps of the tracking stream: kernel trace + HIP API trace of a short bench run (gpurun_out/gaps); tools/gap_trace.py reads them
set -e
O=gpurun_out/gaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/prof -o b -- python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-one-submission-variant --no-local-map-sweep --steps 120 --warmup 30 > $O/bench.log 2>&1
ls -la $O/prof/*/ | head
python3 tools/gap_trace.py $O/prof > $O/gaps.txt
cat $O/gaps.txt
find $O/prof -name "*.csv" -size +20M -delete
