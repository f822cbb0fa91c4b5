"""What the host does in the two gaps of a frame on the tracking stream (between the motion-model stage's solver and the frustum kernel,
between the local-map stage's solver and the next frame's projection): from a rocprofv3 kernel trace + HIP API trace of bench.py, per gap the
HIP calls of the tracking thread with their start offsets (medians over the frames)."""
import csv, glob, sys, collections, statistics
d = sys.argv[1]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
ht = glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)[0]
K = [r for r in csv.DictReader(open(kt))]
H = [r for r in csv.DictReader(open(ht))]
def name(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), name(r)) for r in K))
rp = [(s, e, n) for s, e, n in ks if n.startswith('k_resolve_pose')]
nxt_names = {'k_resolve_pose<1': 'k_project_queries', 'k_resolve_pose<0': 'k_frustum_queries'}
hs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function'], r['Thread_Id']) for r in H))
import bisect
hstart = [h[0] for h in hs]
for key, nxt in nxt_names.items():
    rows = collections.defaultdict(list); gaps = []
    for s, e, n in rp:
        if not n.startswith(key): continue
        i = bisect.bisect_left([k[0] for k in ks], e)
        f = next((k for k in ks[i:i + 40] if k[2].startswith(nxt)), None)
        if not f or f[0] - e > 300000: continue
        gaps.append((f[0] - e) / 1e3)
        a = bisect.bisect_left(hstart, e - 20000); b = bisect.bisect_left(hstart, f[0])
        # the thread that launches the next kernel = the tracking thread
        tid = None
        for h in hs[a:b][::-1]:
            if h[2].startswith('hipLaunchKernel') or h[2].startswith('hipExtLaunch') or h[2] == 'hipModuleLaunchKernel': tid = h[3]; break
        seq = collections.Counter()
        for h in hs[a:b]:
            if h[3] != tid: continue
            seq[h[2]] += 1
            rows[(h[2], seq[h[2]])].append(((h[0] - e) / 1e3, (h[1] - h[0]) / 1e3))
    print(f"gap {key}> end -> {nxt} start: median {statistics.median(gaps):.1f} us over {len(gaps)} frames; HIP calls of the launching thread (start offset from the kernel's end, duration; medians):")
    out = []
    for (fn, k), v in rows.items():
        if len(v) < len(gaps) // 2: continue
        out.append((statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), fn, k))
    for off, dur, fn, k in sorted(out): print(f"   {off:8.1f} us  {dur:7.1f} us  {fn} #{k}")
