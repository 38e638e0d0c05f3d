#!/usr/bin/env python3
"""Per-kernel AND per-grid durations from a rocprofv3 --kernel-trace directory: the --stats summary folds the four ADM
scales (and VIF scales 1-3) of one template instance into one line; this keeps them apart by grid size.
usage: kernel_times.py DIR [--skip N]     (--skip: launches per group to drop as cold, default 1)
Prints median / min / mean microseconds per launch, steady state."""
import argparse, collections, csv, glob, os, statistics

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--skip", type=int, default=1)
a = ap.parse_args()
f = glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
groups = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pqa::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("pqa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
    groups[(name, grid)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print(f"{'kernel':46s} {'grid':>14s} {'n':>4s} {'median us':>10s} {'min us':>9s} {'mean us':>9s}")
for (name, grid), v in sorted(groups.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    d = [x[1] / 1e3 for x in v][a.skip:] or [x[1] / 1e3 for x in v]
    print(f"{name[:46]:46s} {str(grid[0]) + 'x' + str(grid[1]):>14s} {len(d):4d} {statistics.median(d):10.1f} {min(d):9.1f} {statistics.mean(d):9.1f}")
