set -e
O=gpurun_out/r4a; mkdir -p $O
python3 bench.py --cpu-frames 0 --no-lane-variant --steps 300 --warmup 60 > $O/bench_new.json 2> $O/bench_new.err
ASD_RESOLVE=bids python3 bench.py --cpu-frames 0 --no-lane-variant --steps 300 --warmup 60 > $O/bench_old.json 2> $O/bench_old.err
python3 bench.py --cpu-frames 0 --no-lane-variant --steps 300 --warmup 60 > $O/bench_new2.json 2>> $O/bench_new.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 --no-lane-variant > $O/bench_prof.log 2>&1
python3 tools/timeline.py $O/prof > $O/timeline.txt
find $O/prof -name "*kernel_trace.csv" -delete
for f in $O/bench_new.json $O/bench_old.json $O/bench_new2.json; do python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cat $O/timeline.txt
