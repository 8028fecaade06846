#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_step.py into profiles/<name>.json:
HBM-side bytes per launch of the GEMM family (2 x FETCH_SIZE per the gfx950 correction + WRITE_SIZE, counters in KiB) against
the algorithmic bytes of those launches; adamw_kernel is the control (its traffic is known exactly).
usage: python tools/pmc_summary.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, sys
fetch_csv, write_csv, out = sys.argv[1:4]
STEPS = 3


def collect(path, counter):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        fam = "gemm" if ("gemm_pipe" in k or "splitk_reduce" in k) else ("adamw" if "adamw_kernel" in k else None)
        if fam:
            tot[fam] = tot.get(fam, 0.0) + float(r["Counter_Value"])
            n[fam] = n.get(fam, 0) + 1
    return tot, n


f, nf = collect(fetch_csv, "FETCH_SIZE")
w, nw = collect(write_csv, "WRITE_SIZE")
assert nf == nw, (nf, nw)
gemm_bytes = (2 * f["gemm"] + w["gemm"]) * 1024
adam_bytes = (2 * f["adamw"] + w["adamw"]) * 1024
# algorithmic bytes of the GEMM launches of one step (config A, B = 256): every operand and output once
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
alg_step = 4653449216.0       # bench.py gemm accounting (profiles/README.md), bytes per step
adam_alg = 3369384960
res = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domains) -- python3 tools/pmc_step.py: "
              f"{STEPS} eager MAE ViT-B/16 steps at B=256; gfx950 correction: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); counters in KiB",
    "check": {"kernel": "adamw_kernel", "launches": nf["adamw"], "traffic_bytes_per_launch": round(adam_bytes / nf["adamw"]),
              "algorithmic_bytes_per_launch": adam_alg, "ratio": round(adam_bytes / nf["adamw"] / adam_alg, 5)},
    "family": "gemm_pipe_kernel* + gemm_pipe_group_kernel + splitk_reduce_kernel", "launches": nf["gemm"],
    "fetch_size_kib": f["gemm"], "write_size_kib": w["gemm"],
    "traffic_bytes_per_launch": round(gemm_bytes / nf["gemm"]), "traffic_bytes_per_step": round(gemm_bytes / STEPS),
    "algorithmic_bytes_per_launch": alg_step * STEPS / nf["gemm"], "algorithmic_bytes_per_step": alg_step,
    "note": "traffic is counted at the L2s' memory side: every XCD's L2 fetches what its tiles touch (8 L2s, not coherent) and "
            "Infinity-Cache hits are counted too; writes are the outputs plus the split-K slabs",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
