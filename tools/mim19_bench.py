#!/usr/bin/env python3
"""BASELINE configs[4] alone (configs/mim_19.ini: SimMIM ViT-Large/16, 5x128x128, mask ratio 0.6, bs 128, bf16): the leg
bench.py reports under extra.mim_19 -- a clean target for rocprofv3."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.bench_mim19(None, torch.device("cuda", 0))))
