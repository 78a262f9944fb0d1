#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 kernel_stats.csv: tools/kstats.py <file> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]
for r in rows:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>7s} avg_us {float(r["AverageNs"]) / 1e3:8.2f} pct {r["Percentage"]}')
