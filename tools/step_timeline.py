#!/usr/bin/env python
"""One replayed step out of a rocprofv3 --kernel-trace CSV: where the time between two steps' first kernels goes.
    python tools/step_timeline.py trace.csv [step-index-from-the-end] [--all]
Prints the step's span, the busy time per queue, every gap > 2 us on the critical queue, and the kernels < 10 us summed by name."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 2
marks = [i for i, r in enumerate(rows) if "obj_split" in r["Kernel_Name"]]
# whole steps only (bench.py also times the object tower alone at the end): segments that contain the local loss' backward
full = [(a, b) for a, b in zip(marks, marks[1:]) if any("xsoftmax_bwd" in r["Kernel_Name"] for r in rows[a:b])]
# ... and, where there are any, the graph-replayed ones (text tower on its own queue; bench.py's eager timing region runs on one)
multi = [(a, b) for a, b in full if len({r["Queue_Id"] for r in rows[a:b]}) > 1]
a, b = (multi or full)[-min(back, len(multi or full))]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"step span {span:.1f} us, {len(seg)} kernels")
busy = defaultdict(float)
for r in seg:
    busy[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("busy per queue (us):", {k: round(v, 1) for k, v in busy.items()})
# union of busy intervals over all queues -> idle time of the whole device
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
idle, cur_end, gaps = 0.0, iv[0][0], []
for s, e in iv:
    if s > cur_end:
        idle += (s - cur_end) / 1e3
        gaps.append(((cur_end - t0) / 1e3, (s - cur_end) / 1e3))
    cur_end = max(cur_end, e)
print(f"device idle inside the step: {idle:.1f} us in {len(gaps)} gaps; gaps > 2 us:")
byname = {int(r["Start_Timestamp"]): r["Kernel_Name"] for r in seg}
starts = sorted(byname)
for at, g in sorted(gaps, key=lambda x: -x[1])[:25]:
    if g > 2.0:
        nxt = next((byname[s] for s in starts if (s - t0) / 1e3 >= at + g - 0.01), "?")
        print(f"   t = {at:9.1f} us  gap {g:6.1f} us  before {nxt[:70]}")
small = defaultdict(lambda: [0, 0.0])
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 10.0 or "--all" in sys.argv:
        small[r["Kernel_Name"][:90]][0] += 1
        small[r["Kernel_Name"][:90]][1] += d
print("kernels under 10 us (count, total us):")
for k, (n, t) in sorted(small.items(), key=lambda kv: -kv[1][1]):
    print(f"   {n:4d} {t:8.1f}  {k}")
print(f"   total {sum(v[0] for v in small.values())} launches, {sum(v[1] for v in small.values()):.1f} us")
