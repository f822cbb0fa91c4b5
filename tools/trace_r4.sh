# kernel trace of the bench: tracking-chain timeline + the extractor's two queues (gpurun_out/r4a)
set -e
O=gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 150 --warmup 45 > $O/bench_prof.log 2>&1
python3 tools/timeline.py $O/prof > $O/timeline.txt
python3 tools/extractor_timeline.py $O/prof > $O/extractor_timeline.txt
python3 tools/ba_segment.py $O/prof > $O/ba_segment.txt
find $O/prof -name "*kernel_trace.csv" -delete
tail -8 $O/timeline.txt; head -1 $O/extractor_timeline.txt
