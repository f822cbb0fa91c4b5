# kernel-time A/B of LocalBA's kernels between two builds of the library (rocprofv3 --stats on tools/ba_times.py): tools/solve_ab.sh LIB_A LIB_B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf gpurun_out/sab; ASDHIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sab -o s -- python3 tools/ba_times.py > /dev/null 2>&1
  echo "$lib:"; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/sab/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    if n.startswith('k_ba_') and int(r['Calls'])>=30: print(f"   {n:22s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
done
rm -rf gpurun_out/sab
