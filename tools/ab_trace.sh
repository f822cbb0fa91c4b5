# A/B on one box: kernel averages (rocprofv3 --stats) of the bench under different switches (gpurun_out/r4a)
O=gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() {  # $1 = label, rest = env assignments
  lbl=$1; shift
  rm -rf $O/prof_$lbl
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lbl -o b -- python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 150 --warmup 45 > $O/prof_$lbl.log 2>&1 )
  python3 - $O/prof_$lbl $lbl <<'PY'
import csv,glob,sys,re,json
d,lbl=sys.argv[1],sys.argv[2]
f=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return re.sub(r"\(.*$","",n)[:28]
want=("k_window_search","k_resolve2","k_pose_opt","k_frustum","k_project","k_conv_x3<32, 32","k_ba_solve","k_ba_schur")
out=[]
for r in rows:
    n=short(r["Name"])
    if n.startswith(want): out.append(f"{n}={float(r['AverageNs'])/1e3:.1f}")
log=open(d+".log").read().strip().splitlines()
j=[l for l in log if l.startswith("{")]
v=json.loads(j[-1]) if j else {}
print(lbl, round(v.get("value",0),1), v.get("steady_state",{}).get("ms_tracking_per_frame"), " ".join(sorted(out)))
PY
  find $O/prof_$lbl -name "*kernel_trace.csv" -delete
}
stats early X=1
stats inorder ASD_CHAIN_EARLY=0
stats early_q8 GPU_MAX_HW_QUEUES=8
stats inorder_q8 ASD_CHAIN_EARLY=0 GPU_MAX_HW_QUEUES=8
stats early2 X=1
