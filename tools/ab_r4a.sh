# A/B of round 4's tracking-stream changes on one box (gpurun_out/r4a): one submission per frame against two, sorted-list replay against bids
set -e
O=gpurun_out/r4a; mkdir -p $O
run() { python3 bench.py --cpu-frames 0 --no-lane-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "chain       $(run)"
echo "no-chain    $(run --no-chain)"
echo "chain       $(run)"
echo "bids        $(ASD_RESOLVE=bids run --no-chain)"
timeout -k 10 300 python -m pytest tests/test_bench_host.py tests/test_track_chain.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 --no-lane-variant > $O/bench_prof.log 2>&1
python3 tools/timeline.py $O/prof > $O/timeline.txt
find $O/prof -name "*kernel_trace.csv" -delete
cat $O/timeline.txt
