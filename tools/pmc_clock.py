#!/usr/bin/env python3
"""effective clock per kernel from a --pmc GRBM_GUI_ACTIVE pass: cycles (summed over 8 XCDs) / 8 / duration"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = r["Kernel_Name"].split("(")[0][-60:]
    a = acc[k]; a[0] += float(r["Counter_Value"]) / 8; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[2] += 1
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k:62s} n {n:4d} avg {ns/n/1e3:8.1f} us  cycles {cyc/n:10.0f}  clock {cyc/ns:.2f} GHz")
