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
q = [r for r in rows if r["n"].startswith("k_pose_opt")][0]["Queue_Id"]
tr = sorted([r for r in rows if r["Queue_Id"] == q], key=lambda r: r["s"])
starts = [i for i, r in enumerate(tr) if r["n"].startswith("k_window_search")]
frames = starts[0::2]                      # two searches per frame (frame-to-frame, local map)
i0, i1 = frames[-8], frames[-7]
t0, prev = tr[i0]["s"], None
print(f"tracking stream = queue {q}; one frame ({(tr[i1]['s'] - t0) / 1e3:.1f} us):")
for r in tr[i0:i1]:
    gap = (r["s"] - prev) / 1e3 if prev else 0.0
    print(f'{(r["s"] - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(r["e"] - r["s"]) / 1e3:7.1f}  {r["n"]}')
    prev = r["e"]
per, busy = [], []
for a, b in zip(frames[-20:-1], frames[-19:]):
    per.append((tr[b]["s"] - tr[a]["s"]) / 1e3)
    busy.append(sum(r["e"] - r["s"] for r in tr[a:b]) / 1e3)
print(f"last {len(per)} frames: period {sum(per) / len(per):.1f} us, kernels on the tracking stream {sum(busy) / len(busy):.1f} us, gaps {sum(per) / len(per) - sum(busy) / len(busy):.1f} us")
