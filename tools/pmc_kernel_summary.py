#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes (one CSV per pass): launches, matrix-core utilisation, L2 hit rate, HBM-side bytes.
usage: pmc_kernel_summary.py <out.json> "<source note>" <min share of GPU cycles> <pass.csv> [<pass.csv> ...]"""
import collections, csv, json, re, sys
out, note, min_share, paths = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4:]
SIMDS, XCDS = 1024, 8
def short(k):
    k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", k)
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    return k[:90]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for p in paths:
    for r in csv.DictReader(open(p)):
        f, c = short(r["Kernel_Name"]), r["Counter_Name"]
        acc[f][c] += float(r["Counter_Value"]); cnt[f][c] += 1
total = sum(a.get("GRBM_GUI_ACTIVE", 0.0) for a in acc.values()) or 1.0
res = {"source": note,
       "definitions": {"mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)", "l2_hit_rate": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
                       "hbm_side_bytes_per_launch": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / launches (gfx950: FETCH_SIZE counts half, MI355X_MICROARCH.md)"},
       "kernels": {}}
for f in sorted(acc, key=lambda f: -acc[f].get("GRBM_GUI_ACTIVE", 0.0)):
    a = acc[f]
    if a.get("GRBM_GUI_ACTIVE", 0.0) / total < min_share:
        continue
    n = max(cnt[f].values())
    e = {"launches": n, "share_of_gpu_cycles": round(a.get("GRBM_GUI_ACTIVE", 0.0) / total, 4)}
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / XCDS
        e["gpu_cycles_per_launch"] = round(cyc / cnt[f]["GRBM_GUI_ACTIVE"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            e["mfma_util"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS), 4)
    if "TCC_HIT_sum" in a:
        e["l2_hit_rate"] = round(a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0.0)), 4)
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        e["hbm_side_bytes_per_launch"] = round((2 * a["FETCH_SIZE"] / cnt[f]["FETCH_SIZE"] + a["WRITE_SIZE"] / cnt[f]["WRITE_SIZE"]) * 1024)
    res["kernels"][f] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
