#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (steady state: the
last N dispatches).  usage: tools/kt_gaps.py <dir> [N]"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "picsong" in r["Kernel_Name"]][-n:]
dur, gap = defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"][:60]
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[k].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
tot = (int(rows[-1]["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} dispatches over {tot:.1f} us")
for k in dur:
    print(f"  {k:60s} n={len(dur[k]):3d} dur {sum(dur[k]) / len(dur[k]) / 1e3:6.2f} us  gap after {sum(gap[k]) / len(gap[k]) / 1e3:6.2f} us")
