# One gpurun call that produces every artifact of profiles/r05_* (copy them from gpurun_out/r05p afterwards).
# Every step must succeed: a non-zero exit of the profiled process fails the script (round 2 tolerated an exit-time segfault here
# with `|| echo`; its cause -- the CU-masked stream, profiles/r03_teardown_diagnostics.txt -- is gone).
set -e
O=gpurun_out/r05p; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-one-submission-variant --no-local-map-sweep > $O/bench_prof.log 2>&1
test -s $O/prof/bench_kernel_stats.csv
python3 tools/timeline.py $O/prof > $O/timeline.txt
python3 tools/extractor_timeline.py $O/prof > $O/extractor_timeline.txt
python3 tools/ba_segment.py $O/prof > $O/ba_segment.txt
# host view of the default (two-call) step, and the one-submission variant's chain on the device clock (stamps of the kernels themselves)
ASD_TIMING=1 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-one-submission-variant --no-local-map-sweep --steps 400 --warmup 60 2> $O/step_host.err > /dev/null
grep -E "track_loop|search\+resolve" $O/step_host.err | tail -14 > $O/step_host_timers.txt
ASD_TIMING=1 python3 bench.py --chain --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-local-map-sweep --steps 400 --warmup 60 2> $O/chain_clock.err > /dev/null
grep -E "track_frame (resolve|device)" $O/chain_clock.err | tail -3 > $O/chain_clock.txt
# BASELINE configs[3] (stereo) through the C++ host, the driver's K = 20 / W = 5 command, the per-keyframe stage
python3 bench.py --workload euroc-stereo --steps 300 --warmup 45 > $O/bench_euroc_stereo.json 2> $O/bench_euroc_stereo.err
python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 > $O/bench_k20.json 2> $O/bench_k20.err
python3 tools/kf_times.py --reps 12 > $O/kf_ops.json 2> $O/kf_ops.err
python3 bench.py --sequences 11 --seq-scale 0.05 > $O/sequences_1gpu.json 2> $O/sequences_1gpu.err
find $O/prof -name "*kernel_trace.csv" -delete
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1
python3 tools/collect_traffic.py $O/pmc_fetch $O/pmc_write 2000 > $O/traffic_conv2.json
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -o c -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o m -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1
python3 tools/collect_mfma_util.py $O/pmc_clk $O/pmc_mfma > $O/asdnet_mfma_util.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_clk $O/pmc_mfma
python3 tools/time_asdnet.py 2000 20 > $O/time_asdnet.txt 2>&1
ASD_TIMING=1 python3 tools/ba_times.py > $O/ba_times.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ba -o ba -- python3 tools/ba_times.py > /dev/null 2>&1
find $O/prof_ba -name "*kernel_trace.csv" -delete
# what a workgroup of every conv layer spends its life on, at three load levels (in-kernel stamps)
for n in 64 256 2000; do echo "n=$n"; ASD_X3_PHASES=1 python3 tools/x3_clock.py $n 2>&1; done > $O/asdnet_phases.txt
# the experimental LDS-image / weight-ring kernels (asdnet_ring.hip): per-layer times and SQ counters beside the default kernels'
for m in 0 1 2 4 6; do echo "ASD_ASDNET_RING=$m"; ASD_ASDNET_RING=$m python3 tools/time_asdnet.py 2000 20 2>&1 | tail -2; done > $O/asdnet_ring_times.txt
RING=0 bash tools/ring_pmc.sh > $O/asdnet_sq_counters_default.txt 2>&1
RING=6 bash tools/ring_pmc.sh > $O/asdnet_sq_counters_ring.txt 2>&1
rm -rf gpurun_out/ringpmc
# PoseOptimization beside the extractor: one result over thousands of calls
python3 tools/diag/pose_determinism.py 3000 beside > $O/pose_determinism.txt 2>&1
python3 tools/diag/ba_determinism.py 600 >> $O/pose_determinism.txt 2>&1
tail -3 $O/timeline.txt; tail -2 $O/time_asdnet.txt; tail -3 $O/ba_times.txt; tail -2 $O/pose_determinism.txt
