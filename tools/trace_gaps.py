#!/usr/bin/env python3
"""Per-launch durations AND inter-kernel gaps of the replayed step from a rocprofv3 kernel_trace.csv.
usage: trace_gaps.py <kernel_trace.csv> [rows]  -- takes the last 20 steps (each step starts with its mask_kernel launch)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "mask_kernel" in r["Kernel_Name"] and "simmim" not in r["Kernel_Name"]]
first, last = starts[-21], starts[-1]              # 20 whole steps (the noise draw in front of mask_kernel goes with the previous step)
steps = 20
sel = rows[first:last]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / steps / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / steps / 1e3
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(sel[:-1], sel[1:])]
print(f"launches/step {len(sel)/steps:.1f}  wall/step {span:.1f} us  sum of kernel durations {busy:.1f} us  sum of gaps {sum(gaps)/steps/1e3:.1f} us "
      f"(median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, negative (overlap) {sum(1 for g in gaps if g < 0)/steps:.1f}/step)")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", n)
    return n.split("(")[0][:60]
agg = collections.defaultdict(list)
for r in sel:
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
for (name, wgs, lds), v in tot[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    v2 = sorted(v)
    print(f"{sum(v)/steps:8.1f} us/step  n/step {len(v)/steps:5.1f}  med {v2[len(v2)//2]:7.2f}  min {v2[0]:7.2f}  wgs {wgs:5d} lds {lds:6d}  {name}")
