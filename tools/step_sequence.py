#!/usr/bin/env python3
"""The launches of ONE replayed training step in order, from a rocprofv3 kernel_trace.csv: position, kernel, workgroups, LDS,
median duration and median gap to the previous kernel over the last 20 steps (a step starts with its mask_kernel launch).
usage: step_sequence.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "mask_kernel" in r["Kernel_Name"] and "simmim" not in r["Kernel_Name"]]
steps = [rows[a:b] for a, b in zip(starts[-21:-1], starts[-20:])]
n = len(steps[0])
assert all(len(s) == n for s in steps), sorted({len(s) for s in steps})
def short(x):
    x = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", x)
    x = re.sub(r"\(skyemb_gemm_args\)|\(char const\*\)", "", x)
    return x[:78]
med = lambda v: sorted(v)[len(v) // 2]
tot = 0.0
for i in range(n):
    d = med([(int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in steps])
    g = med([(int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"])) / 1e3 for s in steps]) if i else 0.0
    r = steps[-1][i]
    tot += d + g
    print(f"{i:4d} {d:7.2f} us gap {g:5.2f}  t={tot:7.1f}  wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} x{int(r['Workgroup_Size_X']):4d} lds {int(r['LDS_Block_Size']):6d}  {short(r['Kernel_Name'])}")
