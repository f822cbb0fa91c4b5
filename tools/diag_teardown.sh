# ONE-OFF REPRODUCER, not part of any routine or refresh run: it deliberately re-triggers the ROCm 7.2 exit deadlock / rocprofiler SIGSEGV of
# CU-masked streams on shared hardware.  The evidence it produced is kept in profiles/r03_teardown_diagnostics.txt; the library no longer
# creates such a stream.  Refuses to run unless asked explicitly.
if [ "${ASD_DIAG_TEARDOWN_OPT_IN:-0}" != "1" ]; then echo "$0: one-off reproducer (hangs by design); set ASD_DIAG_TEARDOWN_OPT_IN=1 to run it anyway"; exit 0; fi
# Teardown diagnostics (DESIGN.md "Teardown"): what a CU-masked stream does at hipStreamDestroy and at process exit, alone and under
# rocprofv3, without library code (tools/ubench/masked_stream_exit) and with it (bench.py, module map dumped for frame resolution).
# One gpurun call:  bash tools/diag_teardown.sh    -> gpurun_out/td/summary.txt
O=gpurun_out/td; mkdir -p $O; : > $O/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
U=tools/ubench/masked_stream_exit
for m in leak_plain leak_masked destroy_masked leak_masked_reset; do
  timeout -k 5 60 $U $m 1 $O/maps_$m.txt > $O/plain_$m.log 2>&1; echo "plain     $m rc=$?" >> $O/summary.txt
  timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$m -o x -- $U $m 1 $O/maps_prof_$m.txt > $O/prof_$m.log 2>&1; echo "rocprofv3 $m rc=$?" >> $O/summary.txt
done
ASD_DUMP_MAPS=$O/maps_bench.txt timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bench -o b -- python3 bench.py --steps 100 --cpu-frames 0 --no-lane-variant > $O/bench_prof.log 2>&1; echo "rocprofv3 bench.py (masked ASDNet stream) rc=$?" >> $O/summary.txt
ASD_EXTRACT_RESERVE_CUS=0 ASD_DUMP_MAPS=$O/maps_bench_nomask.txt timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bench0 -o b -- python3 bench.py --steps 100 --cpu-frames 0 --no-lane-variant > $O/bench_prof_nomask.log 2>&1; echo "rocprofv3 bench.py ASD_EXTRACT_RESERVE_CUS=0 rc=$?" >> $O/summary.txt
find $O -name "*kernel_trace.csv" -delete
cat $O/summary.txt
