# Copies the summaries tools/refresh_profiles.sh left under gpurun_out/r05p (scratch) into profiles/r05_* (tracked).
set -e
O=gpurun_out/r05p; P=profiles
cp $O/bench.json $P/r05_bench.json; cp $O/bench_k20.json $P/r05_bench_k20_w5.json
cp $O/prof/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
cp $O/timeline.txt $P/r05_tracking_timeline.txt; cp $O/extractor_timeline.txt $P/r05_extractor_timeline.txt; cp $O/ba_segment.txt $P/r05_ba_segment.txt
cp $O/step_host_timers.txt $P/r05_step_host_timers.txt; cp $O/chain_clock.txt $P/r05_chain_device_clock.txt
cp $O/bench_euroc_stereo.json $P/r05_bench_euroc_stereo.json; cp $O/kf_ops.json $P/r05_kf_ops.json; cp $O/sequences_1gpu.json $P/r05_sequences_1gpu.json
cp $O/traffic_conv2.json $P/r05_traffic_conv2.json; cp $O/traffic_conv2.json $P/traffic_conv2.json; cp $O/asdnet_mfma_util.json $P/r05_asdnet_mfma_util.json
cp $O/time_asdnet.txt $P/r05_time_asdnet.txt; cp $O/ba_times.txt $P/r05_ba_times.txt
cp $O/prof_ba/ba_kernel_stats.csv $P/r05_local_ba_kernel_stats.csv
cp $O/asdnet_phases.txt $P/r05_asdnet_phases.txt; cp $O/asdnet_ring_times.txt $P/r05_asdnet_ring_times.txt
cp $O/asdnet_sq_counters_default.txt $P/r05_asdnet_sq_counters_default.txt; cp $O/asdnet_sq_counters_ring.txt $P/r05_asdnet_sq_counters_ring.txt
cp $O/pose_determinism.txt $P/r05_pose_determinism.txt
