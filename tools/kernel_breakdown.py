#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats kernel_stats.csv -> per-batch kernel time table:  python tools/kernel_breakdown.py <csv> <n_batches>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nb = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per batch {tot / nb / 1e3:.1f} us over {len(rows)} kernels")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f"{float(r['TotalDurationNs']) / nb / 1e3:8.2f} us/batch  calls/batch {int(r['Calls']) / nb:5.2f}  avg {float(r['AverageNs']) / 1e3:7.2f} us  {r['Name'][:100]}")
