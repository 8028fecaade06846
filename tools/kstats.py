#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: usage kstats.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/steps/1e6:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step {100*float(r['TotalDurationNs'])/tot:5.1f}% calls/step {int(r['Calls'])/steps:6.1f} "
          f"avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
