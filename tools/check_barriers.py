"""Every s_barrier in the device code must have `s_waitcnt ... lgkmcnt(0)` in front of it in its own basic block with no LDS
instruction, call or label in between (the scheduler may put ALU instructions there): a wave's LDS stores have landed before any
other wave is let past the barrier.  asd_syncthreads() (ctx.h) writes
the wait out; this check (part of `make check-isa`) keeps a barrier from slipping in without it."""
import re
import sys

bad = 0
total = 0
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    func = "?"
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        if not re.match(r"\s+s_barrier\b", l):
            continue
        total += 1
        ok = False
        j = i - 1
        while j >= 0:
            t = lines[j].strip()
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                ok = True
                break
            if re.match(r"^(\.LBB|_Z)", lines[j]) or t.startswith("ds_") or "lds" in t.split(" ")[0] or t.startswith("s_swappc") or t.startswith("s_setpc") or t.startswith("s_barrier"):
                break
            j -= 1
        if not ok:
            bad += 1
            print(f"{path}:{i + 1}: s_barrier without 's_waitcnt lgkmcnt(0)' in front of it, in {func[:90]}")
print(f"check-barriers: {total} barriers, {bad} without the LDS wait")
sys.exit(1 if bad else 0)
