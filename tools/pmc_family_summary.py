#!/usr/bin/env python3
"""Per-kernel-family summary of rocprofv3 --pmc passes over tools/pmc_step.py (three eager ViT-B steps):
matrix-core utilisation (SQ_VALU_MFMA_BUSY_CYCLES against the SIMD-cycles the launches had), LDS bank conflicts, L2 hit rate.
usage: pmc_family_summary.py <out.json> <pass.csv> [<pass.csv> ...]      (every CSV = one pass; counters may repeat)"""
import csv, json, re, sys, collections
out, paths = sys.argv[1], sys.argv[2:]
SIMDS, XCDS = 1024, 8
# executions of the step in the profiled run = launches of a once-per-step kernel (pmc_step.py: three steps + the warm-up
# execution TrainStep makes when the optimiser step is fused into the weight-gradient launches)
_first = list(csv.DictReader(open(paths[0])))
_c0 = _first[0]["Counter_Name"]
STEPS = sum(1 for r in _first if r["Counter_Name"] == _c0 and "loss_finalize" in r["Kernel_Name"])
def family(k):
    if "gemm_pipe_group" in k: return "gemm_group (weight gradients)"
    if "gemm_pipe_kernel" in k:
        m = re.search(r"gemm_pipe_kernel<(\d+), (\d+), (true|false), (true|false), (\d+), (\d+), (\d+), (\d+)>", k)
        return f"gemm {m.group(1)}x{m.group(2)} {'kc' if m.group(3)=='true' else 'rc'}.{'kc' if m.group(4)=='true' else 'rc'} stages {m.group(5)} k-groups {m.group(8)}" if m else "gemm"
    if "splitk_reduce" in k: return "splitk_reduce"
    if "mha_fwd" in k: return "mha_fwd"
    if "mha_bwd" in k: return "mha_bwd"
    if "ln_" in k: return "layernorm"
    if "adamw" in k: return "adamw"
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(lambda: collections.defaultdict(float))
for p in paths:
    for r in csv.DictReader(open(p)):
        f = family(r["Kernel_Name"])
        if f is None: continue
        c = r["Counter_Name"]
        acc[f][c] += float(r["Counter_Value"]); cnt[f][c] += 1
        dur[f][c] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
res = {"source": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/pmc_step.py (three eager MAE ViT-B/16 steps, B = 256; one pass per "
                 "counter group, no tracing domains; tools/pmc_multi.sh), summarised by tools/pmc_family_summary.py",
       "definitions": {"mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of the launches' SIMD-cycles in which a matrix "
                                     "instruction was executing (GRBM_GUI_ACTIVE is summed over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back)",
                       "mfma_busy_cycles_per_step": "SQ_VALU_MFMA_BUSY_CYCLES per step (a v_mfma_f32_16x16x32_bf16 holds its SIMD for 16 cycles of 8 passes: "
                                                     "the GEMMs' 81.5 M wave-instructions per step are the check)",
                       "lds_conflict_share": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "l2_hit_rate": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"},
       "families": {}}
for f in sorted(acc, key=lambda f: -acc[f].get("GRBM_GUI_ACTIVE", 0)):
    a = acc[f]; e = {}
    n = max(cnt[f].values())
    e["launches_per_step"] = round(n / STEPS, 1)
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / XCDS
        e["gpu_cycles_per_step"] = round(cyc / STEPS)
        e["kernel_us_per_step_under_pmc"] = round(dur[f]["GRBM_GUI_ACTIVE"] / STEPS / 1e3, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            e["mfma_busy_cycles_per_step"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / STEPS)
            e["mfma_util"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS), 4)
        if "SQ_BUSY_CYCLES" in a: e["sq_busy_cycles_per_step"] = round(a["SQ_BUSY_CYCLES"] / STEPS)
        if "SQ_WAVES" in a: e["waves_per_step"] = round(a["SQ_WAVES"] / STEPS)
        if "SQ_WAVE_CYCLES" in a: e["wave_cycles_x4_per_step"] = round(a["SQ_WAVE_CYCLES"] / STEPS)
    if "SQ_LDS_IDX_ACTIVE" in a and a["SQ_LDS_IDX_ACTIVE"]:
        e["lds_conflict_share"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"], 4)
    if "TCC_HIT_sum" in a:
        e["l2_hit_rate"] = round(a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0.0)), 4)
    if "FETCH_SIZE" in a: e["hbm_side_read_bytes_per_step"] = round(2 * a["FETCH_SIZE"] * 1024 / STEPS)     # gfx950: FETCH_SIZE counts half
    if "WRITE_SIZE" in a: e["hbm_side_write_bytes_per_step"] = round(a["WRITE_SIZE"] * 1024 / STEPS)
    res["families"][f] = e
g = [v for k, v in res["families"].items() if k.startswith("gemm")]
if g and all("mfma_busy_cycles_per_step" in v for v in g):
    busy, cyc = sum(v["mfma_busy_cycles_per_step"] for v in g), sum(v["gpu_cycles_per_step"] for v in g)
    res["gemm_family_total"] = {"mfma_busy_cycles_per_step": busy, "gpu_cycles_per_step": cyc, "mfma_util": round(busy / (cyc * SIMDS), 4),
                                "expected_busy_cycles_from_flops": round(1.3356e12 / 16384 * 16)}
    # HBM-side traffic of the family (the GEMM launches + the split-K reduce launches that finish them) against its algorithmic bytes:
    # every operand and output once (4.653 GB per step at config A, B = 256: tools/gemm_bench.py shapes) + the optimiser state the fused
    # weight-gradient launches move (round 5: 26 B per parameter in the epilogue form, 34 B as side jobs: 3.04 GB per step at config A)
    fam = [v for k, v in res["families"].items() if k.startswith("gemm") or k == "splitk_reduce"]
    if all("hbm_side_read_bytes_per_step" in v and "hbm_side_write_bytes_per_step" in v for v in fam):
        tot = sum(v["hbm_side_read_bytes_per_step"] + v["hbm_side_write_bytes_per_step"] for v in fam)
        launches = sum(v["launches_per_step"] for v in fam)
        res["gemm_family_total"].update({"hbm_side_bytes_per_step": tot, "launches_per_step": launches,
                                         "hbm_side_bytes_per_launch": round(tot / launches),
                                         "algorithmic_bytes_per_step_gemm_operands": 4653449216,
                                         "optimizer_bytes_in_weight_gradient_epilogues": 3038773248,
                                         "note": "tools/pmc_step.py runs TrainStep(use_graph=False): with one process the AdamW step of the transformer blocks' weights "
                                                 "(110.1 M of 112.3 M parameters) runs in the grouped weight-gradient launches: 26 B per parameter in the epilogue form (the "
                                                 "encoder's twelve blocks and decoder block 0), 34 B per parameter as side jobs (decoder blocks 7..1: gradient stored and read "
                                                 "back) -- inside this family's bytes"})
        if "adamw" in res["families"] and "hbm_side_read_bytes_per_step" in res["families"]["adamw"]:
            a = res["families"]["adamw"]
            res["gemm_family_total"]["adamw_rest"] = {"measured_bytes": a["hbm_side_read_bytes_per_step"] + a["hbm_side_write_bytes_per_step"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
