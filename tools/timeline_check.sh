set -e
O=gpurun_out/r05t; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-one-submission-variant --no-local-map-sweep > $O/bench_prof.log 2>&1
python3 tools/timeline.py $O/prof > $O/timeline.txt
head -32 $O/timeline.txt
find $O/prof -name "*kernel_trace.csv" -delete
