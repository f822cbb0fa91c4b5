# tuning aid: SQ counters of the ring kernels (one rocprofv3 --pmc pass per group), summarised per kernel
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ringpmc; rm -rf $O; mkdir -p $O
export ASD_ASDNET_RING=${RING:-7}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/a -o a -- python3 $R/tools/time_asdnet.py 2000 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $O/b -o b -- python3 $R/tools/time_asdnet.py 2000 3 > $O/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c -o c -- python3 $R/tools/time_asdnet.py 2000 3 > $O/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os, re
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/ringpmc"
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(.*$","",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")).strip()
        if "ring" not in k and "k_conv_x3" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in acc:
    print(k[:70], "avg_us %.1f"%(sum(dur[k])/len(dur[k])/1e3))
    for c,v in sorted(acc[k].items()):
        v=v[len(v)//2:]
        print("   %-28s %.4g"%(c,sum(v)/len(v)))
PY
