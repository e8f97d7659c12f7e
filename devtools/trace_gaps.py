"""Timeline of the training steps in a rocprofv3 kernel trace: per step wall time, kernel-busy time, idle gaps.
A step = from one adam_multi_kernel end to the next."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_multi_kernel")]
if len(adam) < 4:
    sys.exit("fewer than 4 Adam launches in the trace")
lo, hi = adam[-4], adam[-1]          # the last three steps
steps = 3
seg = ev[lo + 1: hi + 1]
wall = seg[-1][1] - ev[lo][1]
busy, gaps, prev_end = 0, [], ev[lo][1]
per = defaultdict(lambda: [0, 0])
for s, e, n in seg:
    busy += e - max(s, prev_end) if e > prev_end else 0
    if s > prev_end:
        gaps.append((s - prev_end, n))
    prev_end = max(prev_end, e)
    k = n.split("(")[0][:60]
    per[k][0] += e - s
    per[k][1] += 1
print(f"per step: wall {wall / steps / 1e3:.1f} us, busy {busy / steps / 1e3:.1f} us, idle {(wall - busy) / steps / 1e3:.1f} us, "
      f"{len(seg) / steps:.0f} launches")
gaps.sort(reverse=True)
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
print("gap histogram: >20us %d, 5-20us %d, <5us %d (all three steps)" % (
    sum(g > 20e3 for g, _ in gaps), sum(5e3 < g <= 20e3 for g, _ in gaps), sum(g <= 5e3 for g, _ in gaps)))
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:62s} {c / steps:5.1f} x {t / c / 1e3:8.1f} us = {t / steps / 1e3:8.1f} us/step")
