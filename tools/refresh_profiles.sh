set -e
O=gpurun_out/r02p; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (the profiled process may end with a segfault AFTER rocprofv3 has written its output: the runtime tears the never-destroyed CU-masked
#  stream down under the tool's intercept at exit -- see DESIGN.md, "A runtime hang found on the way"; the tables are complete)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-frames 0 > $O/bench_prof.log 2>&1 || echo "rocprofv3 run ended with status $?" >> $O/bench_prof.log
test -s $O/prof/bench_kernel_stats.csv
python3 tools/timeline.py $O/prof > $O/timeline.txt
find $O/prof -name "*kernel_trace.csv" -delete
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1 || true
python3 tools/collect_traffic.py $O/pmc_fetch $O/pmc_write 2000 > $O/traffic_conv2.json
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -o c -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o m -- python3 tools/time_asdnet.py 2000 3 > /dev/null 2>&1 || true
python3 tools/collect_mfma_util.py $O/pmc_clk $O/pmc_mfma > $O/asdnet_mfma_util.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_clk $O/pmc_mfma
python3 tools/time_asdnet.py 2000 20 > $O/time_asdnet.txt 2>&1
tail -3 $O/timeline.txt; cat $O/time_asdnet.txt | tail -2
