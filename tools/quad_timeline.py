#!/usr/bin/env python3
"""Timeline of the quad path's kernels from a rocprofv3 --kernel-trace csv: per kernel name calls / avg / min / max, then the last `n` dispatches
(start relative to the first of them, duration, name) -- do the draws of batch k + 1 run next to the word passes of batch k?
Usage: quad_timeline.py <dir with *_kernel_trace.csv> [n]"""
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name.replace("(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
by = {}
for s, e, n, q in rows:
    by.setdefault(n, []).append(e - s)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:42s} calls {len(d):6d}  total {sum(d) / 1e6:9.3f} ms  avg {sum(d) / len(d) / 1e3:9.2f} us  min {min(d) / 1e3:9.2f}  max {max(d) / 1e3:9.2f}")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = rows[-n:]
t0 = tail[0][0]
print()
for s, e, name, q in tail:
    print(f"  +{(s - t0) / 1e3:9.2f} us  .. +{(e - t0) / 1e3:9.2f}  ({(e - s) / 1e3:8.2f} us)  q{q}  {name}")
