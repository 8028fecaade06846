#!/usr/bin/env python3
"""Five Q = 16 searches over a 1 M x 768 bank (k = 100): a small target for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import search
bank = torch.randn(1_000_000, 768, device="cuda"); w = torch.rand(768, device="cuda") + 0.5
pb = search.PreparedBank(bank, w); q = torch.randn(16, 768, device="cuda")
for _ in range(5):
    s, i = search.cosine_topk(q, pb, 100)
torch.cuda.synchronize(); print("done", int(i.sum()))
