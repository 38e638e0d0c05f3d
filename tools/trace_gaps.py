#!/usr/bin/env python3
"""Where a step's wall time goes beyond its kernels: from a rocprofv3 --kernel-trace CSV of bench.py, the span from the
first to the last pqa kernel of each timed step, the sum of kernel durations inside it and the largest idle gaps.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [frames_per_step]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pqa::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("::")[-1].split("(")[0][:40]) for r in rows)
# split into steps at gaps > 300 us (host work between steps)
steps, cur = [], [ev[0]]
for a, b in zip(ev, ev[1:]):
    if b[0] - a[1] > 300_000:
        steps.append(cur); cur = []
    cur.append(b)
steps.append(cur)
for i, s in enumerate(steps):
    span = (s[-1][1] - s[0][0]) / 1e3
    busy = sum(e[1] - e[0] for e in s) / 1e3
    gaps = sorted(((b[0] - a[1]) / 1e3, a[2], b[2]) for a, b in zip(s, s[1:]))[::-1]
    print(f"step {i}: {len(s)} kernels, span {span:.0f} us, kernels {busy:.0f} us, idle inside {span - busy:.0f} us ({100 * (span - busy) / span:.1f} %); "
          f"largest gaps: " + "; ".join(f"{g:.1f} us after {a}" for g, a, b in gaps[:4]))
