#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV: per kernel total time and calls, the union of busy intervals (how much of the wall
window had at least one kernel running), and how much of it had two or more.   python tools/trace_busy.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = defaultdict(lambda: [0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("chaindp::", "")
    per[n][0] += 1
    per[n][1] += (e - s) / 1e6
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy1 = busy2 = 0
depth, last = 0, t0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
print(f"window {(t1 - t0) / 1e6:.1f} ms, >=1 kernel running {busy1 / 1e6:.1f} ms ({100 * busy1 / (t1 - t0):.0f} %), >=2 running {busy2 / 1e6:.1f} ms")
for n, (c, ms) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n[:60]:60s} {c:7d} calls {ms:10.2f} ms total {ms / c * 1e3:10.1f} us avg")
