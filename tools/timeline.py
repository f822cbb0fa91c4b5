"""Timeline of the tracking stream from a rocprofv3 kernel trace of bench.py: kernels, durations and the gaps between them
for one steady-state frame, plus per-frame totals.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 60 --warmup 30 --cpu-frames 0
  python3 tools/timeline.py gpurun_out/trace"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:34]


for r in rows:
    r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
# round 4: the PoseOptimization kernels run on a stream of their own (launched ahead, waiting on a device flag): the tracking chain is
# the search queue plus the solver queue; a solver kernel's span starts when it became resident, not when its inputs were ready
q = [r for r in rows if r["n"].startswith("k_window_search")][0]["Queue_Id"]
qs = {q, [r for r in rows if r["n"].startswith(("k_resolve_pose", "k_pose_opt", "k_track_solver"))][0]["Queue_Id"]}   # (the solver stream of ASD_CHAIN_EARLY, if any)
tr = sorted([r for r in rows if r["Queue_Id"] in qs], key=lambda r: r["s"])
starts = [i for i, r in enumerate(tr) if r["n"].startswith("k_window_search")]
frames = starts[0::2]                      # two searches per frame (frame-to-frame, local map)
i0, i1 = frames[-8], frames[-7]
t0, prev = tr[i0]["s"], None
print(f"tracking chain = queues {sorted(qs)}; one frame ({(tr[i1]['s'] - t0) / 1e3:.1f} us):")
for r in tr[i0:i1]:
    gap = (r["s"] - prev) / 1e3 if prev else 0.0
    print(f'{(r["s"] - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(r["e"] - r["s"]) / 1e3:7.1f}  {r["n"]}')
    prev = r["e"]
per, busy = [], []
for a, b in zip(frames[-20:-1], frames[-19:]):
    per.append((tr[b]["s"] - tr[a]["s"]) / 1e3)
    busy.append(sum(r["e"] - r["s"] for r in tr[a:b]) / 1e3)
print(f"last {len(per)} frames: period {sum(per) / len(per):.1f} us, kernels on the tracking stream {sum(busy) / len(busy):.1f} us, gaps {sum(per) / len(per) - sum(busy) / len(busy):.1f} us")

# frames that ran beside a LocalBA (any k_ba_* kernel of another queue inside the frame's span) against those that did not
ba = sorted((r["s"], r["e"]) for r in rows if r["n"].startswith("k_ba_"))
if ba:
    import bisect
    bs = [b[0] for b in ba]
    with_ba, without = [], []
    for a, b in zip(frames[2:-1], frames[3:]):
        t_a, t_b = tr[a]["s"], tr[b]["s"]
        k = bisect.bisect_left(bs, t_a)
        n_ba = 0
        while k < len(ba) and ba[k][0] < t_b:
            n_ba += 1
            k += 1
        (with_ba if n_ba else without).append(((t_b - t_a) / 1e3, n_ba, sum(r["e"] - r["s"] for r in tr[a:b]) / 1e3, (a, b)))
    if with_ba and without:
        med = lambda v: sorted(v)[len(v) // 2]
        print(f"frames beside LocalBA kernels: {len(with_ba)}, median period {med([p for p, _, _, _ in with_ba]):.1f} us "
              f"(tracking kernels {med([k for _, _, k, _ in with_ba]):.1f} us, {sum(n for _, n, _, _ in with_ba) / len(with_ba):.0f} BA kernels inside); "
              f"other frames: {len(without)}, median period {med([p for p, _, _, _ in without]):.1f} us "
              f"(tracking kernels {med([k for _, _, k, _ in without]):.1f} us)")
        names = sorted({r["n"] for r in tr})
        for nm in names:
            a = [r["e"] - r["s"] for _, _, _, (x, y) in with_ba for r in tr[x:y] if r["n"] == nm]
            b = [r["e"] - r["s"] for _, _, _, (x, y) in without for r in tr[x:y] if r["n"] == nm]
            if a and b:
                print(f"   {nm:36s} beside BA {sum(a) / len(a) / 1e3:7.1f} us x{len(a) / len(with_ba):.1f}   alone {sum(b) / len(b) / 1e3:7.1f} us x{len(b) / len(without):.1f}")

# dispatch waits: time between the end of the previous kernel of the same queue and the start of the big single-workgroup kernels
# (they need most of a CU; beside the extractor's ASDNet workgroups that takes a while)
def med(v):
    return sorted(v)[len(v) // 2] if v else float("nan")
byq = {}
for r in sorted(rows, key=lambda r: r["s"]):
    byq.setdefault(r["Queue_Id"], []).append(r)
for name in ("k_ba_solve_lds", "k_ba_schur", "k_ba_step", "k_ba_linearize", "k_resolve_pose", "k_resolve2", "k_window_search"):
    gaps, durs = [], []
    for qq, lst in byq.items():
        for a, b in zip(lst[:-1], lst[1:]):
            if b["n"].startswith(name):
                gaps.append((b["s"] - a["e"]) / 1e3)
                durs.append((b["e"] - b["s"]) / 1e3)
    if gaps:
        print(f"   {name:18s} x{len(gaps):5d}: gap in front (median / mean) {med(gaps):6.1f} / {sum(gaps) / len(gaps):6.1f} us, duration median {med(durs):6.1f} us")
