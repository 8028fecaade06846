#!/bin/bash
# HBM-side traffic of the Q=16 bank pass (round 6): FETCH_SIZE and WRITE_SIZE in separate passes over tools/search_small.py
set -e
bash tools/pmc_multi.sh r06spmc fetch "FETCH_SIZE" tools/search_small.py
bash tools/pmc_multi.sh r06spmc write "WRITE_SIZE" tools/search_small.py
python3 - <<'PY'
import csv, json
def tot(path, counter):
    v, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "cosine_topk_stream_kernel" in r["Kernel_Name"]:
            v += float(r["Counter_Value"]); n += 1
    return v, n
f, nf = tot("gpurun_out/r06spmc/pmc_fetch.csv", "FETCH_SIZE")
w, nw = tot("gpurun_out/r06spmc/pmc_write.csv", "WRITE_SIZE")
assert nf == nw and nf > 0
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domains) -- python3 tools/search_small.py "
                 "(Q = 16 searches over 1 M x 768, k = 100, with the sample floor); gfx950 correction: FETCH_SIZE doubled; counters in KiB",
       "kernels": {"cosine_topk_stream_kernel<8>": {"launches": nf, "fetch_size_kib_per_launch": f / nf, "write_size_kib_per_launch": w / nw,
                   "traffic_bytes_per_launch": (2 * f / nf + w / nw) * 1024,
                   "algorithmic_bytes_per_launch": 1000000 * 768 * 4 + 1000000 * 4 + 16 * 768 * 4 + 16 * 100 * 12}}}
json.dump(res, open("gpurun_out/r06spmc/r06_topk_stream_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
python3 tools/fingerprint.py gpurun_out/r06spmc/r06_topk_stream_pmc.json
