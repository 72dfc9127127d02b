#!/usr/bin/env python3
"""rocprofv3 --kernel-trace kernel_trace.csv -> duration of every call of one kernel (substring), in launch order:
    python tools/kernel_calls.py <kernel_trace.csv> <substring> [max_rows]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[:n]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:9.2f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}  wg {r.get('Workgroup_Size_X', '?')}")
