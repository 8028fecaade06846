#!/usr/bin/env python3
"""Copies the achieved-error record the GPU parity tests write (gpurun_out/parity_errors.json: tests/helpers.record_parity) into
profiles/<name>.json with the bars the tests enforce beside it.  usage: python tests/parity_report.py profiles/r03_parity_errors.json"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "parity_errors.json")
data = json.load(open(src))
out = {"source": "python -m pytest tests -m gpu on one MI355X: every figure is |HIP path - oracle| as the named test computed it "
                 "(tests/test_mae_parity_gpu.py::test_full_size_config_a_against_oracle, tests/test_simmim_parity_gpu.py::"
                 "test_mim19_geometry_against_oracle, tests/test_ddp_gpu.py); the oracle is oracle/mae_oracle.py (fp32, CPU), pinned by the "
                 "reference-made goldens", "north_star_bar": "1e-3 relative (fp32 parity mode)", "achieved": data}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
