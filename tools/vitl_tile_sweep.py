#!/usr/bin/env python3
"""Every product tile code on the eight ViT-L block shapes of mim_19 with the step's fused epilogues (us per launch, 20 launches per
HIP graph): is the plan's / the tuned table's choice still the fastest?  usage: vitl_tile_sweep.py [bf16|f16] [vitb]
(vitb: the encoder / decoder block shapes of config A instead: 1280 x 768-wide and 4352 x 512-wide)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd.ops import ACT_DGELU, ACT_GELU, KC, RC
M, D = 8320, 1024
dev = "cuda"
T = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
TILES = (0, 64064, 128064, 128128, 2256128, 6128064, 6064064, 9064064, 9128128, 9144064, 13144256, 256256, -1)
VITB = len(sys.argv) > 2 and sys.argv[2] == "vitb"


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / 100 * 1e3


ws = torch.zeros(8 * 1024 * 1024, device=dev)
# (round 6: half a second of matrix work before the first timing -- the first group measured after start-up read 10-15 % slow:
# qkv fwd 77.4 us first, 65.7 us as the last of its group, same launch)
_wa, _wb = torch.randn(4096, 4096, device=dev).to(T), torch.randn(4096, 4096, device=dev).to(T)
for _ in range(400):
    torch.matmul(_wa, _wb)
torch.cuda.synchronize()
print(f"# {T}; us per launch; tile 0 = what the step launches", flush=True)
SHAPES = [(M, n, N, K, epi) for n, N, K, epi in (("qkv", 3 * D, D, "bias"), ("proj", D, D, "resid"), ("fc1", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "resid"))]
if VITB:
    SHAPES = [(m, f"{tag}.{n}", N, K, epi) for tag, m, d in (("enc", 1280, 768), ("dec", 4352, 512))
              for n, N, K, epi in (("qkv", 3 * d, d, "bias"), ("proj", d, d, "resid"), ("fc1", 4 * d, d, "gelu"), ("fc2", d, 4 * d, "resid"))]
for M, name, N, K, epi in SHAPES:
    x = torch.randn(M, K, device=dev).to(T)
    w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    dy = torch.randn(M, N, device=dev).to(T)
    bias = torch.zeros(N, device=dev)
    y, y2 = torch.empty(M, N, device=dev, dtype=T), torch.empty(M, N, device=dev, dtype=T)
    y32, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
    dx, aux = torch.empty(M, K, device=dev, dtype=T), torch.randn(M, K, device=dev).to(T)
    for tag in ("fwd", "dgrad"):
        out = []
        for tile_ in TILES:
            tile = max(tile_, 0)                       # (-1: the step's launch once more, at the END of the group -- order effects)
            if tag == "fwd":
                if epi == "gelu":
                    f = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ACT_GELU, out=y, out2=y2, ws=ws, tile=tile)
                elif epi == "resid":
                    f = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, ws=ws, tile=tile)
                else:
                    f = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y, ws=ws, tile=tile)
            else:
                kw = dict(act=ACT_DGELU, aux=aux, ldaux=K) if name.endswith("fc2") else {}
                f = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, ws=ws, tile=tile, **kw)
            try:
                out.append((timeit(f), tile_))
            except Exception as e:
                torch.cuda.synchronize()
                out.append((float("inf"), tile_))
        t0 = out[0][0]
        t0 = min(t0, out[-1][0])
        best = min(out[1:-1])
        print(f"{name:8s} {tag:5s} step {out[0][0]:6.1f} / {out[-1][0]:6.1f} (first / last) | " + "  ".join(f"{tile}:{t:.1f}" for t, tile in sorted(out[1:-1])[:5]) +
              ("   <-- %.1f %% faster" % (100 * (1 - best[0] / t0)) if best[0] < 0.97 * t0 else ""), flush=True)
