#!/usr/bin/env python
"""Host input feed for SEVERAL ranks on one host: N concurrent copies of
tools/input_bench.py -- each with its own RegionBatcher, pinned double buffer, staging threads and copy stream -- against the one GPU of this
box, for N = 1, 2, 4, 8.  What it measures: how the HOST side (staging memcpys into pinned memory, thread pools, the Python loop) scales when
eight loaders share the host's cores and memory controllers; all copies share ONE PCIe link here (8 on the real node), so the aggregate is a
lower bound of what eight ranks can feed.  A rank needs 8.0 GB/s at 3.5 k pairs/s (2.37 MB of raw region features per pair).
    python tools/input_feed_scaling.py [threads-per-loader ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
threads = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
ncpu = len(os.sched_getaffinity(0))
print(f"host cores available to this process: {ncpu}")
for n in (1, 2, 4, 8):
    for t in threads:
        if n * t > 2 * ncpu:
            continue
        env = dict(os.environ, STAGE_WORKERS=str(t))
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "input_bench.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(n)]
        rates = []
        for p in procs:
            out = p.communicate()[0]
            m = re.search(r"([0-9.]+) GB/s staged", out)
            rates.append(float(m.group(1)) if m else 0.0)
        print(f"{n} loaders x {t} staging threads: aggregate {sum(rates):6.2f} GB/s  (per loader min {min(rates):.2f} max {max(rates):.2f}; a rank needs 8.0)", flush=True)
