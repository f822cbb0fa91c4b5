"""The extractor's two queues in a rocprofv3 kernel trace (bench.py or tools/extract_throughput.py): period of the ASDNet queue, its
kernel time, the idle time between one forward's k_l2norm and the next forward's first conv, a listing of both queues over two
frames, and the distance from a frame's last front-half kernel (k_angle_patch) to its first ASDNet kernel.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 150 --warmup 30 --cpu-frames 0
  python3 tools/extractor_timeline.py gpurun_out/trace"""
import csv,re,statistics as st,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return re.sub(r"\(.*$","",n)[:30]
for r in rows:
    r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['n']=short(r['Kernel_Name'])
qx=[r for r in rows if r['n'].startswith('k_l2norm')][-1]['Queue_Id']
qf=[r for r in rows if r['n'].startswith('k_fast_score')][-1]['Queue_Id']
X=sorted([r for r in rows if r['Queue_Id']==qx], key=lambda r:r['s'])
F=sorted([r for r in rows if r['Queue_Id']==qf], key=lambda r:r['s'])
starts=[i for i,r in enumerate(X) if r['n'].startswith('k_conv_x3<32, 32, 32')]
per=[];busy=[];gap=[];inner=[]
for a,b in zip(starts[10:-1],starts[11:]):
    per.append((X[b]['s']-X[a]['s'])/1e3); busy.append(sum(r['e']-r['s'] for r in X[a:b])/1e3)
    l2=[r for r in X[a:b] if r['n'].startswith('k_l2norm')][0]
    gap.append((X[b]['s']-l2['e'])/1e3)
    inner.append((l2['e']-X[a]['s'])/1e3 - sum(r['e']-r['s'] for r in X[a:b] if r['s']<=l2['s'])/1e3)
print('asdnet queue: period %.1f  kernels %.1f  l2norm-end -> next conv2 %.1f  gaps inside a forward %.1f'%(st.median(per),st.median(busy),st.median(gap),st.median(inner)))
a=starts[20]
t0=X[a]['s']
t1=X[starts[22]]['s']
for r in sorted([r for r in rows if t0<=r['s']<=t1 and r['Queue_Id'] in (qx,qf)], key=lambda r:r['s']): print(f"   {(r['s']-t0)/1e3:8.1f} {(r['e']-r['s'])/1e3:7.1f} q{r['Queue_Id']} {r['n']}")
# front half: angle_patch end (last front kernel) of frame i vs conv2 start of same frame
ap=[r for r in F if r['n'].startswith('k_angle_patch')]
c2=[X[i] for i in starts]
d=[]
for c in c2[10:-1]:
    prev=[x for x in ap if x['e']<=c['s']]
    if prev: d.append((c['s']-prev[-1]['e'])/1e3)
print('angle_patch end -> conv2 start (same frame): median %.1f us'%st.median(d))
