"""One LocalBA inside the pipelined bench, from a rocprofv3 kernel trace: every kernel of the tracking queue between the first
k_ba_struct_flags of a run and the next k_project_queries, with the gap in front of it; sums per kernel name."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:30]
for r in rows:
    r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
q = [r for r in rows if r["n"].startswith("k_ba_solve")][0]["Queue_Id"]
tr = sorted([r for r in rows if r["Queue_Id"] == q], key=lambda r: r["s"])
starts = [i for i, r in enumerate(tr) if r["n"].startswith("k_ba_struct_flags")]
i0 = starts[len(starts) // 2]
i1 = next(i for i in range(i0, len(tr)) if tr[i]["n"].startswith("k_project_queries"))
prev_end = tr[i0 - 1]["e"]
print(f"LocalBA segment: {i1 - i0} kernels, {(tr[i1]['s'] - tr[i0 - 1]['e']) / 1e3:.1f} us from the end of the kernel in front of it to the next frame's first kernel")
tot_gap = 0
agg = {}
for k, r in enumerate(tr[i0:i1]):
    gap = (r["s"] - prev_end) / 1e3
    tot_gap += max(gap, 0)
    a = agg.setdefault(r["n"], [0, 0.0, 0.0]); a[0] += 1; a[1] += (r["e"] - r["s"]) / 1e3; a[2] += max(gap, 0)
    if k < 40 or gap > 20:
        print(f"{(r['s'] - tr[i0]['s']) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(r['e'] - r['s']) / 1e3:7.1f}  {r['n']}")
    prev_end = max(prev_end, r["e"])
print(f"kernel time {sum(a[1] for a in agg.values()):.1f} us, gaps {tot_gap:.1f} us")
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"   {n:32s} x{a[0]:3d}  {a[1]:8.1f} us  gaps in front {a[2]:7.1f}")
