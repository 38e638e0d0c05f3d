#!/usr/bin/env python3
"""Per-kernel SQ counter summary from two rocprofv3 --pmc passes (profiles/*_sq_counters.txt).

    rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \\
        SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d A -- python3 bench.py ...
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \\
        SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d B -- ...
    python tools/sq_counters.py A B > profiles/<tag>_sq_counters.txt

SQ_*CYCLES counters tick in quad-cycles (x4 = shader clocks).  'valu_act' is the time a wave spends issuing VALU
instructions; valu_act / wave life x resident waves per SIMD = how busy the SIMD's VALU port is."""
import collections
import csv
import glob
import os
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pqa::" not in k:
                continue
            k = k.replace("pqa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                n[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, n, dur


args = [x for x in sys.argv[1:] if not x.startswith("--json")]
json_out = next((x.split("=", 1)[1] for x in sys.argv[1:] if x.startswith("--json=")), None)
a, na, dur = load(args[0])
b, nb, _ = load(args[1])
summary = {}
print("# rocprofv3 --pmc SQ_* (two passes); cycles = counter x 4 (SQ_*CYCLES count quad-cycles);")
print("# 'valu_act % of wave life' x waves per SIMD = VALU busy share")
for k in sorted(a, key=lambda k: -dur[k]):
    w = a[k]["SQ_WAVES"] or 1.0
    life = 4 * a[k]["SQ_WAVE_CYCLES"] / w
    act = 4 * a[k]["SQ_ACTIVE_INST_VALU"] / w
    wb = b[k]["SQ_WAVES"] if b[k].get("SQ_WAVES") else w * nb[k] / max(na[k], 1)
    wait = 4 * b[k]["SQ_WAIT_ANY"] / max(wb, 1.0)
    conf = b[k]["SQ_LDS_BANK_CONFLICT"] / max(b[k]["SQ_LDS_IDX_ACTIVE"], 1.0)
    # effective shader clock: GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
    gui = b[k].get("GRBM_GUI_ACTIVE", 0.0)
    clk = gui / 8.0 / max(nb[k], 1) / max(dur[k] / na[k] * 1e-6, 1e-12) / 1e9 if gui else 0.0
    summary[k] = {"valu_per_wave": a[k]["SQ_INSTS_VALU"] / w, "salu_per_wave": a[k]["SQ_INSTS_SALU"] / w,
                  "waves_per_launch": w / max(na[k], 1), "us_per_launch": dur[k] / na[k],
                  "lds_conflict_frac": conf, "valu_active_frac_of_wave_life": act / max(life, 1), "clock_ghz": clk}
    print(f"{k:46s} us/launch={dur[k] / na[k]:8.1f} VALU/w {a[k]['SQ_INSTS_VALU'] / w:6.0f} SALU {a[k]['SQ_INSTS_SALU'] / w:5.0f} "
          f"LDS {a[k]['SQ_INSTS_LDS'] / w:5.0f} life {life:7.0f} valu_act {act:6.0f} ({100 * act / max(life, 1):3.0f}% of wave life) "
          f"wait_any {100 * wait / max(life, 1):3.0f}% lds_conflict/active {100 * conf:4.1f}%")
if json_out:
    import json
    json.dump(summary, open(json_out, "w"), indent=1)
