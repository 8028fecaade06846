"""GPU parity tests, one HIP kernel family at a time, through the C ABI (libskyemb.so via ctypes).

Each kernel is compared with the CPU oracle piece it replaces (oracle/mae_oracle.py,
oracle/topk_oracle.c) or, for the floating-point building blocks the oracle delegates to torch
(LayerNorm, attention core, GEMM), with a plain torch fp32 CPU statement of the same op.
Tolerances: bit-exact for index / mask / top-k work; fp32 kernels 1e-5..1e-4 relative; bf16 mode
is compared against the fp32 result of the bf16-rounded operands (accumulation is fp32).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mae_oracle as mo
from oracle import similarity_oracle as so


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from sky_embeddings_amd import ops as _ops
    _ops.lib()  # fail loudly if libskyemb.so is missing
    return _ops


DEV = "cuda"


def dev(t, dtype=None):
    t = t.to(DEV)
    return t.to(dtype) if dtype is not None else t


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(20, 72, 40), (68, 96, 256), (200, 136, 72), (128, 128, 64), (257, 520, 264)])
def test_gemm_layouts(ops, dtype, layouts, tile, shape):
    M, N, K = shape
    a_l, b_l = layouts
    if a_l:  # the contiguous extent of an RC operand (a model dimension in real use) is a multiple of 8
        M = (M + 7) // 8 * 8
    if b_l:
        N = (N + 7) // 8 * 8
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    # asymmetric integer-ish content catches transposed fragments exactly
    A[: min(M, 16), : min(K, 16)] = torch.arange(min(M, 16) * min(K, 16)).reshape(min(M, 16), -1).float() % 7 - 3
    Ar, Br = A.to(dtype).float(), B.to(dtype).float()
    ref = Ar.double() @ Br.double().T
    Ad = dev(A.T.contiguous() if a_l else A, dtype)
    Bd = dev(B.T.contiguous() if b_l else B, dtype)
    out = torch.full((M, N), float("nan"), device=DEV)
    cs = torch.full((M,), float("nan"), device=DEV) if a_l else None  # fused bias-gradient column sum
    ops.gemm(Ad, Bd, M=M, N=N, K=K, a_layout=a_l, b_layout=b_l, out_f32=out, tile=tile, colsum_a=cs)
    torch.cuda.synchronize()
    if cs is not None:
        assert float((cs.cpu().double() - Ar.double().sum(1)).abs().max()) <= 1e-5 * float(Ar.abs().sum(1).max())
    scale = float((Ar.abs().double() @ Br.abs().double().T).max())
    assert float((out.cpu().double() - ref).abs().max()) <= 2e-6 * scale, (dtype, layouts, tile, shape)


@pytest.mark.parametrize("b_l", [0, 1])
@pytest.mark.parametrize("tile", [9064064, 0, 9144064, 13144256])
@pytest.mark.parametrize("shape", [(200, 136, 256), (1280, 768, 768), (70, 264, 384), (64, 64, 128), (407, 520, 128)])
def test_gemm_two_k_group_tile(ops, b_l, tile, shape):
    """The 64x64 tile with two wave groups over k (launches of <= 256 tiles; picked by the heuristic when tile = 0) and the tiles
    whose row STRIDE is below their height (a 144-row image stepping by 136 rows x 64 columns on twelve waves in two k-groups; by
    130 rows x 256 columns on twelve waves: ViT-L's 8320 = 64 x 130 token rows): both operand classes with a k-contiguous A, every
    fused epilogue, ragged tile edges (row counts that are and are not multiples of the stride) -- against fp64."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + K + b_l)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    A[: 16, : 16] = torch.arange(256).reshape(16, 16).float() % 7 - 3
    bias, table = torch.randn(N, generator=g), torch.randn(9, N, generator=g)
    tab_row = torch.randint(0, 9, (M,), generator=g, dtype=torch.int32)
    perm = torch.randperm(M + 5, generator=g)[:M].to(torch.int32)
    perm[3] = -1
    resid = torch.randn(M + 5, N, generator=g)
    T = torch.bfloat16
    Ar, Br = A.to(T).double(), B.to(T).double()
    Ad, Bd = dev(A, T), dev(B.T.contiguous() if b_l else B, T)
    kw = dict(M=M, N=N, K=K, a_layout=0, b_layout=b_l, tile=tile)
    prod = Ar @ Br.T
    scale = float((Ar.abs() @ Br.abs().T).max())
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Bd, out_f32=out, **kw)
    assert float((out.cpu().double() - prod).abs().max()) <= 2e-6 * scale
    out32 = torch.zeros(M + 5, N, device=DEV)
    outlp = torch.zeros(M + 5, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, alpha=0.5, bias=dev(bias), table=dev(table), tab_row=dev(tab_row), ldt=N, dst_row=dev(perm), resid=dev(resid),
             ldr=N, out_f32=out32, out=outlp, **kw)
    ref = torch.zeros(M + 5, N, dtype=torch.float64)
    base = prod * 0.5 + bias.double() + table[tab_row.long()].double()
    ok = perm >= 0
    ref[perm[ok].long()] = base[ok] + resid[perm[ok].long()].double()
    assert float((out32.cpu().double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((outlp.float().cpu().double() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    act, pre = torch.zeros(M, N, device=DEV, dtype=T), torch.zeros(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, bias=dev(bias), act=ops.ACT_GELU, out=act, out2=pre, **kw)
    v = (prod + bias.double()).float()
    assert float((pre.float().cpu() - v).abs().max()) <= 1e-2 * float(v.abs().max())
    assert float((act.float().cpu() - torch.nn.functional.gelu(v)).abs().max()) <= 1e-2 * float(v.abs().max())
    aux = torch.randn(M, N, generator=g)
    auxr = aux.to(T).float().requires_grad_(True)
    torch.nn.functional.gelu(auxr).sum().backward()
    dg = torch.zeros(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, aux=dev(aux, T), ldaux=N, act=ops.ACT_DGELU, out=dg, **kw)
    refd = prod.float() * auxr.grad
    assert float((dg.float().cpu() - refd).abs().max()) <= 1e-2 * float(refd.abs().max())
    # an odd number of k-tiles is refused by the explicit tile and served by the one-group tile when the choice is left open
    if tile in (9064064, 9144064):
        with pytest.raises(Exception):
            ops.gemm(Ad[:, :64].contiguous(), dev(B[:, :64].contiguous(), T), M=M, N=N, K=64, out_f32=out, tile=tile)


@pytest.mark.parametrize("tile", [9064064, 9128128])
def test_gemm_two_k_group_weight_gradients(ops, tile):
    """Row-contiguous operands (weight-gradient class) on the two-k-group tiles: products and the fused bias-gradient column
    sums against fp64; four problems as ONE grouped launch give the bits of four single launches on the same tile."""
    g = torch.Generator().manual_seed(tile % 1000)
    tokens = 384
    shapes = [(192, 768), (768, 192), (200, 136), (576, 264)]       # (N_out, K_in); ragged tile edges
    args, outs, refs, keep = [], [], [], []
    for n_out, k_in in shapes:
        dy32, x32 = torch.randn(tokens, n_out, generator=g), torch.randn(tokens, k_in, generator=g)
        dy, x = dev(dy32, torch.bfloat16), dev(x32, torch.bfloat16)
        dw1, db1 = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
        kw = dict(M=n_out, N=k_in, K=tokens, a_layout=ops.RC, b_layout=ops.RC, lda=n_out, ldb=k_in)
        ops.gemm(dy, x, out_f32=dw1, colsum_a=db1, split_k=1, tile=tile, **kw)
        dyr, xr = dy.double().cpu(), x.double().cpu()
        ref = dyr.T @ xr
        assert float((dw1.cpu().double() - ref).abs().max()) <= 2e-6 * float((dyr.abs().T @ xr.abs()).max())
        assert float((db1.cpu().double() - dyr.sum(0)).abs().max()) <= 1e-5 * float(dyr.abs().sum(0).max())
        dw, db = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
        args.append(ops.gemm_args(dy, x, out_f32=dw, colsum_a=db, **kw))
        keep.append((dy, x))
        outs.append((dw, db))
        refs.append((dw1, db1))
    if tile == 9128128:
        grp = ops.GemmGroup(args, DEV, tile=tile)
        assert grp.ok and grp.total_blocks % 8 == 0
        grp.launch()
        torch.cuda.synchronize()
        for (dw, db), (dw1, db1) in zip(outs, refs):
            assert torch.equal(dw, dw1) and torch.equal(db, db1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 70, 136, 96
    g = torch.Generator().manual_seed(11)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, table = torch.randn(N, generator=g), torch.randn(9, N, generator=g)
    tab_row = torch.randint(0, 9, (M,), generator=g, dtype=torch.int32)
    perm = torch.randperm(M + 5, generator=g)[:M].to(torch.int32)
    perm[3] = -1  # skipped row
    resid = torch.randn(M + 5, N, generator=g)
    Ar, Br = A.to(dtype).float(), B.to(dtype).float()
    base = Ar @ Br.T * 0.5 + bias + table[tab_row.long()]
    tol = 3e-2 if dtype != torch.float32 else 1e-4
    # scatter + table + resid, fp32 out and low-precision out
    out32 = torch.zeros(M + 5, N, device=DEV)
    outlp = torch.zeros(M + 5, N, device=DEV, dtype=dtype)
    ops.gemm(dev(A, dtype), dev(B, dtype), M=M, N=N, K=K, alpha=0.5, bias=dev(bias), table=dev(table), tab_row=dev(tab_row),
             ldt=N, dst_row=dev(perm), resid=dev(resid), ldr=N, out_f32=out32, out=outlp)
    ref = torch.zeros(M + 5, N)
    for m in range(M):
        if perm[m] >= 0:
            ref[perm[m]] = base[m] + resid[perm[m]]
    assert float((out32.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((outlp.float().cpu() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    # GELU: out = gelu(v), out2 = v
    act = torch.zeros(M, N, device=DEV, dtype=dtype)
    pre = torch.zeros(M, N, device=DEV, dtype=dtype)
    ops.gemm(dev(A, dtype), dev(B, dtype), M=M, N=N, K=K, bias=dev(bias), act=ops.ACT_GELU, out=act, out2=pre)
    v = Ar @ Br.T + bias
    assert float((pre.float().cpu() - v).abs().max()) < tol * float(v.abs().max())
    assert float((act.float().cpu() - torch.nn.functional.gelu(v)).abs().max()) < tol * float(v.abs().max())
    # dGELU: out = v * gelu'(aux)
    aux = torch.randn(M, N, generator=g)
    auxr = aux.to(dtype).float().requires_grad_(True)
    torch.nn.functional.gelu(auxr).sum().backward()
    dg = torch.zeros(M, N, device=DEV, dtype=dtype)
    ops.gemm(dev(A, dtype), dev(B, dtype), M=M, N=N, K=K, aux=dev(aux, dtype), ldaux=N, act=ops.ACT_DGELU, out=dg)
    refd = (Ar @ Br.T) * auxr.grad
    assert float((dg.float().cpu() - refd).abs().max()) < tol * float(refd.abs().max())


@pytest.mark.parametrize("tile", [0, 64064, 128064, 6128064, 128128, 9064064, 9144064, 13144256, 2256128])
@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1)])
def test_gemm_prefetch_hint_changes_no_result(ops, tile, layouts):
    """skyemb_gemm_args.prefetch: the launch's workgroups touch a range a LATER launch will read (one 4-byte LDS-DMA read per 128-byte
    line, requested before their first operand loads and landing where the wave's own first operand piece lands after it).  A hint:
    the output must equal the launch without it bit for bit -- every ring tile, every operand class, ranges of 4 bytes, of a ragged
    number of 8 KiB chunks, and far more chunks than the launch has waves (only the first two per wave are taken)."""
    a_l, b_l = layouts
    if a_l == 1 and tile in (9144064, 13144256, 2256128, 9064064):
        pytest.skip("k-contiguous A only")
    M, N, K = 407, 520, 256
    g = torch.Generator().manual_seed(tile % 1000 + 7 * a_l + b_l)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    T = torch.bfloat16
    Ad = dev(A.T.contiguous() if a_l else A, T)
    Bd = dev(B.T.contiguous() if b_l else B, T)
    if a_l:                                                  # row-contiguous operands: whole 8-row chunks
        M, N = 400, 520
        Ad, Bd = dev(A[:M].T.contiguous(), T), dev(B.T.contiguous() if b_l else B, T)
    kw = dict(M=M, N=N, K=K, a_layout=a_l, b_layout=b_l, tile=tile, lda=M if a_l else K, ldb=N if b_l else K)
    bias = dev(torch.randn(N, generator=g))
    ref = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Bd, out_f32=ref, bias=bias, **kw)
    pool = torch.randn(6 * 1024 * 1024, device=DEV)          # 24 MB the hints point into
    for nbytes in (4, 8192 * 3 + 132, 24 * 1024 * 1024):
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(Ad, Bd, out_f32=out, bias=bias, prefetch=pool[: nbytes // 4], **kw)
        assert torch.equal(out, ref), (tile, layouts, nbytes)
    # the end of an allocation: the last chunk's lanes are clamped inside the range
    tail = torch.randn(2048 + 33, device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Bd, out_f32=out, bias=bias, prefetch=tail, **kw)
    assert torch.equal(out, ref)
    # (these launches leave most workgroup slots of the device free, so the hint rode in workgroups of its own behind the tiles:
    # prefetch_job; the next test fills the device, which leaves the hint with the tiles' waves)


@pytest.mark.parametrize("b_l", [0, 1])
def test_gemm_prefetch_hint_carried_by_the_tiles_own_waves(ops, b_l):
    """A launch with more tiles than the device has workgroup slots (1536 tiles of 64 x 64, 768 slots): no workgroups of its own for the
    hint, every wave requests its share ahead of its first operand loads (gemm_pipe_body).  Same bits as without the hint, for a range
    that gives every wave two requests, one, or none."""
    M, N, K = 2048, 3072, 192
    g = torch.Generator().manual_seed(91 + b_l)
    T = torch.bfloat16
    Ad = dev(torch.randn(M, K, generator=g), T)
    B = torch.randn(N, K, generator=g) * 0.2
    Bd = dev(B.T.contiguous() if b_l else B, T)
    kw = dict(M=M, N=N, K=K, a_layout=0, b_layout=b_l, tile=64064)
    ref = torch.empty(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, out=ref, **kw)
    pool = torch.randn(32 * 1024 * 1024, device=DEV)         # 128 MB
    for nbytes in (128 * 1024 * 1024, 40 * 1024 * 1024 + 4, 8192 * 5):
        out = torch.zeros(M, N, device=DEV, dtype=T)
        ops.gemm(Ad, Bd, out=out, prefetch=pool[: nbytes // 4], **kw)
        assert torch.equal(out, ref), nbytes


def test_gemm_group_equals_single_launches(ops):
    """Four wgrad-shaped problems (different output sizes, shared contraction length) as ONE grouped launch give the
    same bits as four single launches without split-K; a problem outside the bf16 subset makes the plan refuse."""
    g = torch.Generator().manual_seed(11)
    tokens = 320
    shapes = [(192, 768), (768, 192), (192, 192), (576, 200)]       # (N_out, K_in); 200 -> ragged 64-tile edge
    args, outs, refs = [], [], []
    keep = []
    for n_out, k_in in shapes:
        dy = dev(torch.randn(tokens, n_out, generator=g), torch.bfloat16)
        x = dev(torch.randn(tokens, k_in, generator=g), torch.bfloat16)
        dw, db = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
        dw1, db1 = torch.empty(n_out, k_in, device=DEV), torch.empty(n_out, device=DEV)
        kw = dict(M=n_out, N=k_in, K=tokens, a_layout=ops.RC, b_layout=ops.RC, lda=n_out, ldb=k_in)
        ops.gemm(dy, x, out_f32=dw1, colsum_a=db1, split_k=1, **kw)
        args.append(ops.gemm_args(dy, x, out_f32=dw, colsum_a=db, **kw))
        keep.append((dy, x))
        outs.append((dw, db))
        refs.append((dw1, db1))
    grp = ops.GemmGroup(args, DEV)
    assert grp.ok and grp.total_blocks % 8 == 0
    grp.launch()
    grp.launch()      # replayable
    for (dw, db), (dw1, db1) in zip(outs, refs):
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
    # the same launch with a prefetch hint on its first problem (what the engine's grouped launches carry): same bits
    hint = torch.randn(1 << 20, device=DEV)
    args[0].prefetch, args[0].prefetch_bytes = hint.data_ptr(), hint.numel() * 4 - 12
    for dw, db in outs:
        dw.fill_(float("nan"))
        db.fill_(float("nan"))
    grp2 = ops.GemmGroup(args, DEV)
    grp2.launch()
    for (dw, db), (dw1, db1) in zip(outs, refs):
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
    bad = ops.gemm_args(keep[0][0].float(), keep[0][1].float(), M=192, N=768, K=tokens, a_layout=ops.RC, b_layout=ops.RC,
                        lda=192, ldb=768, out_f32=outs[0][0])
    assert not ops.GemmGroup([args[1], bad], DEV).ok


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(20, 64), (1280, 768), (37, 512), (9, 1280), (5, 192), (8320, 1024), (8197, 256), (4352, 512)])
def test_layernorm(ops, dtype, shape):
    M, D = shape
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 2 + 0.3
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    dy = torch.randn(M, D, generator=g)
    dyr = dy.to(dtype).float()
    y.backward(dyr)
    yd = torch.empty(M, D, device=DEV, dtype=dtype)
    y32 = torch.empty(M, D, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), yd, mean, rstd, M, D, 1e-6, y32=y32)
    assert relerr(y32, y.detach()) < 2e-6
    assert relerr(yd.float(), y.detach()) < (5e-3 if dtype != torch.float32 else 2e-6)
    nblk = ops.layernorm_bwd_blocks(M)
    part = torch.empty(2, nblk, D, device=DEV)
    g_in = torch.randn(M, D, generator=g)
    g_out = dev(g_in.clone())
    g_lp = torch.empty(M, D, device=DEV, dtype=dtype)
    code = ops.dtype_code(dtype)
    dgam, dbet = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    ops.layernorm_bwd(dev(dy, dtype), dev(x), dev(gamma), mean, rstd, g_out, g_out, g_lp, part, dgam, dbet, M, D, code)
    assert relerr(g_out, g_in + xr.grad) < 5e-6
    assert relerr(g_lp.float(), g_in + xr.grad) < (5e-3 if dtype != torch.float32 else 5e-6)
    assert relerr(dgam, gr.grad) < 1e-5 and relerr(dbet, br.grad) < 1e-5


def test_layernorm_bwd_batched_reduce(ops):
    """dgamma/dbeta of several LayerNorms (different M, D) finished by ONE batched launch == the per-LN reduce."""
    g = torch.Generator().manual_seed(5)
    code = ops.dtype_code(torch.bfloat16)
    entries, direct = [], []
    for M, D in ((4352, 512), (1280, 768), (37, 192)):
        x = dev(torch.randn(M, D, generator=g))
        gamma = dev(torch.randn(D, generator=g))
        dy = dev(torch.randn(M, D, generator=g), torch.bfloat16)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        yd = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
        ops.layernorm_fwd(x, gamma, dev(torch.zeros(D)), yd, mean, rstd, M, D, 1e-6)
        nblk = ops.layernorm_bwd_blocks(M)
        outs = []
        for batched in (False, True):
            part = torch.empty(2, nblk, D, device=DEV)
            gout, dgam, dbet = torch.empty(M, D, device=DEV), torch.full((D,), float("nan"), device=DEV), torch.full((D,), float("nan"), device=DEV)
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, gout, None, part, None if batched else dgam,
                              None if batched else dbet, M, D, code)
            outs.append((part, dgam, dbet, nblk, D))
        direct.append(outs[0])
        entries.append(outs[1])
    # a single-vector item of another width (dbeta = NULL: any [nblk, D] stack of partial rows -> [D])
    parts = dev(torch.randn(6, 3072, generator=g))
    vec = torch.full((3072,), float("nan"), device=DEV)
    entries.append((parts, vec, None, 6, 3072))
    items = ops.ln_reduce_items(entries, DEV)
    ops.layernorm_bwd_reduce_batch(items, 1, 2)      # a sub-range first, then everything
    assert torch.equal(entries[1][1], direct[1][1]) and torch.equal(entries[2][2], direct[2][2])
    assert bool(torch.isnan(entries[0][1]).all()) and bool(torch.isnan(vec).all())
    ops.layernorm_bwd_reduce_batch(items, 0, 4)
    for e, d in zip(entries[:3], direct):
        assert torch.equal(e[1], d[1]) and torch.equal(e[2], d[2])
    ref = parts[0].clone()
    for r in range(1, 6):
        ref += parts[r]                                # the kernel's order for nblk <= 32: one row per row group, summed 0, 1, 2, ...
    assert torch.equal(vec, ref)


@pytest.mark.parametrize("shape", [(320, 192, 192, 768, 0), (1088, 512, 512, 2048, 0), (1280, 768, 768, 3072, 0), (4352, 512, 2048, 512, 128064),
                                   (1280, 768, 3072, 768, 128128), (2112, 1024, 1024, 1024, 256256), (333, 192, 768, 192, 0)])
def test_layernorm_backward_as_a_side_job_of_a_grouped_launch(ops, shape):
    """skyemb_gemm_group_attach_ln_bwd: a LayerNorm backward riding in a grouped weight-gradient launch as side workgroups (a
    block's norm1: it does not depend on the launch's tiles) == skyemb_layernorm_bwd as its own launch, bit for bit -- residual
    gradient, its bf16 copy and the dgamma / dbeta partial-sum table -- on every tile shape that carries it (64 x 64: one
    four-wave block per side workgroup; 128 x 64 / 128 x 128 / 256 x 256: two), rows from 192 to 1024 wide, a row count that is
    not a multiple of the block size; the weight gradient of the same launch is unaffected."""
    from sky_embeddings_amd._lib import BF16, RC
    M, D, n_out, k_in, tile = shape
    g = torch.Generator().manual_seed(M + D)
    x = dev(torch.randn(M, D, generator=g))
    dy = dev(torch.randn(M, D, generator=g), torch.bfloat16)
    gam = dev(1 + 0.1 * torch.randn(D, generator=g))
    g_in = dev(torch.randn(M, D, generator=g))
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, gam, dev(torch.zeros(D)), torch.empty(M, D, device=DEV, dtype=torch.bfloat16), mean, rstd, M, D, 1e-6)
    nb = ops.layernorm_bwd_blocks(M)
    go_a, glp_a, part_a = g_in.clone(), torch.empty(M, D, device=DEV, dtype=torch.bfloat16), torch.full((2, nb, D), float("nan"), device=DEV)
    ops.layernorm_bwd(dy, x, gam, mean, rstd, go_a, go_a, glp_a, part_a, None, None, M, D, BF16)
    T = (M + 63) // 64 * 64                                   # token rows of the weight gradient (a multiple of the k-tile)
    dyw = dev(torch.randn(T, n_out, generator=g), torch.bfloat16)
    xw = dev(torch.randn(T, k_in, generator=g), torch.bfloat16)
    dW, db = torch.empty(n_out, k_in, device=DEV), torch.empty(n_out, device=DEV)
    go_b, glp_b, part_b = g_in.clone(), torch.empty(M, D, device=DEV, dtype=torch.bfloat16), torch.full((2, nb, D), float("nan"), device=DEV)
    grp = ops.GemmGroup([ops.gemm_args(dyw, xw, M=n_out, N=k_in, K=T, a_layout=RC, b_layout=RC, lda=n_out, ldb=k_in, out_f32=dW, colsum_a=db)], DEV,
                        tile=tile, ln_bwd=dict(dy=dy, x=x, gamma=gam, mean=mean, rstd=rstd, g_in=go_b, g_out=go_b, g_lp=glp_b, part=part_b, M=M, D=D))
    assert grp.ok and grp.ln_side and (tile == 0 or grp.info.tile == tile) and grp.total_blocks > grp.tile_blocks
    grp.launch()
    torch.cuda.synchronize()
    assert torch.equal(go_a, go_b) and torch.equal(glp_a, glp_b) and torch.equal(part_a, part_b)
    assert relerr(dW, dyw.float().t() @ xw.float()) < 2e-6 and relerr(db, dyw.float().sum(0)) < 1e-5
    # wider rows than the instance carries are declined by the plan, not mis-served: the caller launches the LayerNorm itself
    if grp.info.tile != 256256:
        wide = dict(dy=torch.empty(8, 1024, device=DEV, dtype=torch.bfloat16), x=torch.empty(8, 1024, device=DEV), gamma=torch.empty(1024, device=DEV),
                    mean=mean, rstd=rstd, g_in=None, g_out=torch.empty(8, 1024, device=DEV), g_lp=None, part=torch.empty(2, 2, 1024, device=DEV), M=8, D=1024)
        g2 = ops.GemmGroup([ops.gemm_args(dyw, xw, M=n_out, N=k_in, K=T, a_layout=RC, b_layout=RC, lda=n_out, ldb=k_in, out_f32=dW, colsum_a=db)], DEV,
                           tile=grp.info.tile, ln_bwd=wide)
        assert g2.ok and not g2.ln_side and g2.total_blocks == g2.tile_blocks


def test_grouped_wgrad_256_tile(ops):
    """The 256 x 256 weight-gradient tile (both operands row-contiguous; csrc/gemm_pipe256.h) in a grouped launch: gradients and
    bias gradients against torch and against the 128 x 128 tile; with the optimiser step in its
    epilogue against the separate AdamW launch on the same gradient."""
    from sky_embeddings_amd._lib import RC, AdamwDesc
    g = torch.Generator().manual_seed(23)
    T = 704                                              # token rows: 11 k-tiles
    shapes = [(512, 256), (256, 768)]                    # dW [n_out, k_in]
    dys = [dev(torch.randn(T, o, generator=g), torch.bfloat16) for o, _ in shapes]
    xs = [dev(torch.randn(T, i, generator=g), torch.bfloat16) for _, i in shapes]
    sizes = [o * i for o, i in shapes]
    offs = [0, sizes[0]]
    n = sum(sizes)

    def problems(flat, dbs=None):
        out = []
        for j, (o, i) in enumerate(shapes):
            kw = {}
            if dbs is not None:
                kw["colsum_a"] = dbs[j]
            out.append(ops.gemm_args(dys[j], xs[j], M=o, N=i, K=T, a_layout=RC, b_layout=RC, lda=o, ldb=i,
                                     out_f32=flat[offs[j]:offs[j] + sizes[j]].view(o, i), **kw))
        return out

    res = {}
    for tile in (128128, 256256):
        flat = torch.full((n,), float("nan"), device=DEV)
        dbs = [torch.full((o,), float("nan"), device=DEV) for o, _ in shapes]
        grp = ops.GemmGroup(problems(flat, dbs), DEV, tile=tile)
        assert grp.ok and grp.info.tile == tile
        grp.launch()
        res[tile] = (flat, dbs)
    for j, (o, i) in enumerate(shapes):
        ref = dys[j].float().t() @ xs[j].float()
        got = res[256256][0][offs[j]:offs[j] + sizes[j]].view(o, i)
        assert relerr(got, ref) < 2e-6
        assert relerr(got, res[128128][0][offs[j]:offs[j] + sizes[j]].view(o, i)) < 2e-6
        assert relerr(res[256256][1][j], dys[j].float().sum(0)) < 1e-5
    # optimiser step in the epilogue
    step, lr, wd, n_decay = 3, 1e-3, 0.05, sizes[0] + 1000
    p0, m0, v0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 0.01
    hyper = dev(torch.tensor([lr, 1 - 0.9 ** step, 1 - 0.95 ** step, 0.0]))
    # ... reference: the gradient of the plain launch through the separate AdamW launch
    pr, mr, vr, gr = dev(p0.clone()), dev(m0.clone()), dev(v0.clone()), res[256256][0].clone()
    plr = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    ops.adamw(pr, gr, mr, vr, plr, n, n_decay, hyper, 0.9, 0.95, 1e-8, wd, grad_scale=0.5)
    pd, md, vd, gd = dev(p0.clone()), dev(m0.clone()), dev(v0.clone()), torch.zeros(n, device=DEV)
    pl = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    d = AdamwDesc()
    d.g_base, d.p, d.m, d.v, d.p_lp, d.hyper = (t.data_ptr() for t in (gd, pd, md, vd, pl, hyper))
    d.n_decay, d.beta1, d.beta2, d.eps, d.weight_decay, d.grad_scale = n_decay, 0.9, 0.95, 1e-8, wd, 0.5
    dbs = [torch.full((o,), float("nan"), device=DEV) for o, _ in shapes]
    grp = ops.GemmGroup(problems(gd, dbs), DEV, tile=256256, adamw=d)
    assert grp.ok and grp.info.tile == 256256
    grp.launch()
    assert torch.equal(pd, pr) and torch.equal(md, mr) and torch.equal(vd, vr) and torch.equal(pl, plr)
    assert bool((gd == 0).all())                          # the gradient buffer is not written
    for j in range(2):
        assert torch.equal(dbs[j], res[256256][1][j])


def test_vit_l_weight_gradient_group_takes_the_256_tile(ops):
    """The four weight gradients of a ViT-L block with the tile choice left open (mim_19's shapes, 2112 token rows here): 192 whole
    256 x 256 tiles, one round -> the plan picks the 256 x 256 group; gradients and bias gradients == the 128 x 128 group's."""
    from sky_embeddings_amd._lib import RC
    g = torch.Generator().manual_seed(41)
    T, dim, hid = 2112, 1024, 4096
    shapes = [(dim, hid), (hid, dim), (dim, dim), (3 * dim, dim)]
    dys = [dev(torch.randn(T, o, generator=g), torch.bfloat16) for o, _ in shapes]
    xs = [dev(torch.randn(T, i, generator=g), torch.bfloat16) for _, i in shapes]
    res = {}
    for tile in (0, 128128):
        dWs = [torch.full((o, i), float("nan"), device=DEV) for o, i in shapes]
        dbs = [torch.full((o,), float("nan"), device=DEV) for o, _ in shapes]
        grp = ops.GemmGroup([ops.gemm_args(dys[j], xs[j], M=o, N=i, K=T, a_layout=RC, b_layout=RC, lda=o, ldb=i, out_f32=dWs[j], colsum_a=dbs[j])
                             for j, (o, i) in enumerate(shapes)], DEV, tile=tile)
        assert grp.ok
        grp.launch()
        res[tile] = (grp.info.tile, grp.total_blocks, dWs, dbs)
    assert res[0][0] == 256256 and res[0][1] == 192 and res[128128][0] == 128128
    for j in range(4):
        assert relerr(res[0][2][j], res[128128][2][j]) < 2e-6 and relerr(res[0][3][j], res[128128][3][j]) < 1e-5
        assert relerr(res[0][2][j], dys[j].float().t() @ xs[j].float()) < 2e-6


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(3, 5, 12, 64), (2, 17, 16, 32), (2, 66, 2, 64), (4, 5, 4, 16), (1, 65, 3, 8),
                                 (2, 32, 3, 32), (5, 16, 2, 64), (3, 1, 2, 32), (7, 31, 5, 64),
                                 (2, 65, 4, 32), (1, 100, 2, 32), (3, 33, 2, 32), (2, 97, 3, 32), (2, 64, 2, 64),
                                 (8, 5, 3, 64), (7, 3, 2, 32), (13, 5, 2, 32), (9, 2, 1, 64), (4, 11, 2, 32), (3, 10, 3, 64)])
def test_attention(ops, dtype, cfg):
    B, N, H, hd = cfg
    D = H * hd
    g = torch.Generator().manual_seed(N * 100 + hd)
    qkv = torch.randn(B, N, 3 * D, generator=g)
    dout = torch.randn(B, N, D, generator=g)
    q_r = qkv.to(dtype).float().requires_grad_(True)
    t = q_r.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = ((t[0] * hd ** -0.5) @ t[1].transpose(-2, -1)).softmax(-1)
    o = (att @ t[2]).transpose(1, 2).reshape(B, N, D)
    o.backward(dout.to(dtype).float())
    out = torch.empty(B, N, D, device=DEV, dtype=dtype)
    dqkv = torch.empty(B, N, 3 * D, device=DEV, dtype=dtype)
    ops.mha_fwd(dev(qkv, dtype), out, B, N, H, hd)
    ops.mha_bwd(dev(qkv, dtype), dev(dout, dtype), dqkv, B, N, H, hd)
    tol = 6e-3 if dtype != torch.float32 else 3e-6
    assert relerr(out.float(), o.detach()) < tol
    assert relerr(dqkv.float(), q_r.grad) < tol


@pytest.mark.parametrize("N", [128, 113, 97])
def test_attention_mfma_strips_full_width(ops, N):
    """97..128 tokens (four 32-token strips per head; the backward launch needs > 64 KB of dynamic LDS), bf16 MFMA path only
    (the fp32 LDS kernel stops below that)."""
    B, H, hd = 2, 3, 64
    D = H * hd
    g = torch.Generator().manual_seed(9)
    qkv, dout = torch.randn(B, N, 3 * D, generator=g), torch.randn(B, N, D, generator=g)
    q_r = qkv.bfloat16().float().requires_grad_(True)
    t = q_r.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    o = (((t[0] * hd ** -0.5) @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(B, N, D)
    o.backward(dout.bfloat16().float())
    out = torch.empty(B, N, D, device=DEV, dtype=torch.bfloat16)
    dqkv = torch.empty(B, N, 3 * D, device=DEV, dtype=torch.bfloat16)
    ops.mha_fwd(dev(qkv, torch.bfloat16), out, B, N, H, hd)
    ops.mha_bwd(dev(qkv, torch.bfloat16), dev(dout, torch.bfloat16), dqkv, B, N, H, hd)
    assert relerr(out.float(), o.detach()) < 6e-3 and relerr(dqkv.float(), q_r.grad) < 6e-3


# ------------------------------------------------------------------------------------ front end
@pytest.mark.parametrize("L,ratio", [(16, 0.75), (64, 0.6), (16, 0.0), (256, 0.9)])
def test_random_mask_from_noise(ops, L, ratio):
    B = 7
    g = torch.Generator().manual_seed(L)
    noise = torch.rand(B, L, generator=g)
    noise[2, 3] = noise[2, 1]  # exact tie -> lower index first
    keep = int(L * (1 - ratio))
    tok = torch.arange(B * L, dtype=torch.float32).reshape(B, L, 1)
    xm, mask_ref, ids_ref = mo.random_masking_from_noise(tok, ratio, noise)
    ids = torch.empty(B, L, dtype=torch.int64, device=DEV)
    mask = torch.empty(B, L, device=DEV)
    ids_keep = torch.empty(B, keep, dtype=torch.int32, device=DEV)
    dd = torch.empty(B, keep + 1, dtype=torch.int32, device=DEV)
    dt = torch.empty(B, keep + 1, dtype=torch.int32, device=DEV)
    ops.random_mask_from_noise(dev(noise), keep, ids, mask, ids_keep, dd, dt)
    assert torch.equal(ids.cpu(), ids_ref) and torch.equal(mask.cpu(), mask_ref)
    assert torch.equal(ids_keep.cpu().long(), (xm[:, :, 0] - torch.arange(B)[:, None] * L).long())
    assert torch.equal(dt.cpu()[:, 1:].long(), ids_keep.cpu().long() + 1) and bool((dt.cpu()[:, 0] == 0).all())
    assert torch.equal(dd.cpu().long(), dt.cpu().long() + torch.arange(B)[:, None] * (L + 1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(5, 64, 16), (9, 64, 8), (3, 32, 4)])
def test_patch_gather_and_pmv_grad(ops, dtype, geom):
    C, H, p = geom
    B, L = 3, (H // p) ** 2
    keep = max(1, L // 4)
    cfg = mo.MAEConfig(img_size=H, patch_size=p, in_chans=C, pixel_mean=0.2, pixel_std=1.7)
    g = torch.Generator().manual_seed(C + p)
    x = torch.randn(B, C, H, H, generator=g)
    x[0, 1] = float("nan")
    x[1, 0, 3:9, 2:30] = float("nan")
    pmv = torch.randn(C, p, p, generator=g)
    ids_keep = torch.stack([torch.randperm(L, generator=g)[:keep] for _ in range(B)]).to(torch.int32)
    xn = mo.norm_inputs(x, cfg)
    xn = torch.where(torch.isnan(xn), pmv.repeat(1, H // p, H // p).expand(B, -1, -1, -1), xn)
    # conv-weight order (c, py, px)
    pat = xn.reshape(B, C, H // p, p, H // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, L, C * p * p)
    ref = torch.gather(pat, 1, ids_keep.long()[:, :, None].expand(-1, -1, C * p * p)).reshape(B * keep, -1)
    out = torch.empty(B * keep, C * p * p, device=DEV, dtype=dtype)
    ops.patch_gather(dev(x), dev(pmv), dev(ids_keep), out, p, keep, 0.2, 1.7)
    assert torch.equal(out.float().cpu(), ref.to(dtype).float())
    # all patches in order
    out_all = torch.empty(B * L, C * p * p, device=DEV, dtype=dtype)
    ops.patch_gather(dev(x), dev(pmv), None, out_all, p, L, 0.2, 1.7)
    assert torch.equal(out_all.float().cpu(), pat.reshape(B * L, -1).to(dtype).float())
    # d patch_mask_values = sum of row-gradients at NaN pixels
    drows = torch.randn(B * keep, C * p * p, generator=g)
    nanpat = torch.isnan(x).float().reshape(B, C, H // p, p, H // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, L, -1)
    nansel = torch.gather(nanpat, 1, ids_keep.long()[:, :, None].expand(-1, -1, C * p * p)).reshape(B * keep, -1)
    ref_d = (drows * nansel).sum(0).reshape(C, p, p)
    part = torch.empty(B, C * p * p, device=DEV)
    dpmv = torch.empty(C, p, p, device=DEV)
    ops.patch_gather_bwd_pmv(dev(x), dev(ids_keep), dev(drows), part, dpmv, p, keep)
    assert float((dpmv.cpu() - ref_d).abs().max()) < 1e-5 * max(1.0, float(ref_d.abs().max()))


def test_glue_kernels(ops):
    g = torch.Generator().manual_seed(3)
    B, L, Dd = 5, 16, 32
    x = torch.randn(B, L + 1, Dd, generator=g)
    mask = (torch.rand(B, L, generator=g) > 0.4).float()
    mt, pos = torch.randn(Dd, generator=g), torch.randn(L + 1, Dd, generator=g)
    xd = dev(x.clone())
    ops.fill_mask_tokens(xd, dev(mask), dev(mt), dev(pos), B, L, Dd)
    ref = x.clone()
    ref[:, 1:][mask.bool()] = (mt + pos[1:]).expand(B, -1, -1)[mask.bool()]
    assert torch.equal(xd.cpu(), ref)
    idx = torch.randint(0, B * (L + 1), (23,), generator=g, dtype=torch.int32)
    o32, olp = torch.empty(23, Dd, device=DEV), torch.empty(23, Dd, device=DEV, dtype=torch.bfloat16)
    ops.gather_rows(dev(x.reshape(-1, Dd)), dev(idx), o32, olp, 23, Dd)
    assert torch.equal(o32.cpu(), x.reshape(-1, Dd)[idx.long()])
    assert torch.equal(olp.cpu(), x.reshape(-1, Dd)[idx.long()].bfloat16())
    part, out = torch.empty(256, Dd, device=DEV), torch.empty(Dd, device=DEV)
    ops.rowsum_select(dev(x.reshape(-1, Dd)), Dd, dev(mask.reshape(-1)), 1, L, L + 1, B * L, Dd, part, out)
    assert float((out.cpu() - (x[:, 1:] * mask[:, :, None]).sum((0, 1))).abs().max()) < 1e-5
    ops.rowsum_select(dev(x.reshape(-1, Dd)), Dd, None, 0, 1, L + 1, B, Dd, part, out)
    assert float((out.cpu() - x[:, 0].sum(0)).abs().max()) < 1e-5
    X = torch.randn(301, 200, generator=g)
    cs = torch.empty(200, device=DEV)
    ops.colsum(dev(X), 301, 200, cs)
    assert float((cs.cpu() - X.sum(0)).abs().max()) < 1e-4
    ops.colsum(dev(X, torch.bfloat16), 301, 200, cs)
    assert float((cs.cpu() - X.bfloat16().float().sum(0)).abs().max()) < 1e-4
    dst = torch.empty(301 * 200, device=DEV, dtype=torch.bfloat16)
    ops.cast(dev(X.reshape(-1)), dst, 301 * 200)
    assert torch.equal(dst.cpu(), X.reshape(-1).bfloat16())


# ------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("norm_pix,loss_fn", [(True, "mse"), (False, "mse"), (True, "L1")])
@pytest.mark.parametrize("geom", [(5, 64, 16), (9, 32, 8)])
def test_masked_patch_loss(ops, norm_pix, loss_fn, geom):
    C, H, p = geom
    B, L, pv = 4, (H // p) ** 2, C * p * p
    cfg = mo.MAEConfig(img_size=H, patch_size=p, in_chans=C, pixel_mean=0.1, pixel_std=1.3, norm_pix_loss=norm_pix,
                       loss_fn=loss_fn)
    g = torch.Generator().manual_seed(H + C)
    x = torch.randn(B, C, H, H, generator=g)
    x[1, 2] = float("nan")
    x[2, 0, 3:9, 5:20] = float("nan")
    mask = (torch.rand(B, L, generator=g) < 0.7).float()
    pred_full = torch.randn(B, L + 1, pv, generator=g)
    pr = pred_full[:, 1:].clone().requires_grad_(True)
    loss_ref = mo.forward_loss(mo.norm_inputs(x, cfg), pr, mask, cfg, nan_safe=True)
    loss_ref.backward()
    # faithful (not nan-safe) forward value is identical
    assert float(mo.forward_loss(mo.norm_inputs(x, cfg), pr.detach(), mask, cfg)) == float(loss_ref)
    loss = torch.empty(1, device=DEV)
    ws = torch.empty(4 * B * L + 4, device=DEV)
    d32 = torch.full((B, L + 1, pv), float("nan"), device=DEV)
    dlp = torch.empty(B, L + 1, pv, device=DEV, dtype=torch.bfloat16)
    first = None
    for _ in range(2):                                      # replays give the same bits
        loss.fill_(float("nan"))
        ops.masked_patch_loss(dev(x), dev(pred_full), dev(mask), loss, dlp, d32, ops.BF16, ws, p, 1, 0.1, 1.3, norm_pix,
                              loss_fn != "mse")
        torch.cuda.synchronize()
        first = float(loss) if first is None else first
        assert float(loss) == first
    assert abs(float(loss) - float(loss_ref)) < 2e-6 * abs(float(loss_ref))
    assert bool((d32[:, 0] == 0).all())
    assert float((d32[:, 1:].cpu() - pr.grad).abs().max()) < 1e-5 * float(pr.grad.abs().max())
    assert relerr(dlp[:, 1:].float(), pr.grad) < 5e-3


# ------------------------------------------------------------------------------------ optimiser
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_adamw(ops, dtype):
    n, n_decay = 4096 + 8, 1000
    g = torch.Generator().manual_seed(5)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    m, v = torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 0.01
    pd, gd, md, vd = dev(p.clone()), dev(gr.clone()), dev(m.clone()), dev(v.clone())
    plp = torch.empty(n, device=DEV, dtype=dtype)
    step, lr, wd = 3, 1e-3, 0.05
    if dtype == torch.float32:  # step scalars from device memory (graph mode) ...
        hyper = dev(torch.tensor([lr, 1 - 0.9 ** step, 1 - 0.95 ** step, 0.0]))
        ops.adamw(pd, gd, md, vd, plp, n, n_decay, hyper, 0.9, 0.95, 1e-8, wd, grad_scale=0.5, zero_grad=True)
    else:  # ... or by value
        ops.adamw(pd, gd, md, vd, plp, n, n_decay, None, 0.9, 0.95, 1e-8, wd, grad_scale=0.5, zero_grad=True, lr=lr,
                  bc1=1 - 0.9 ** step, bc2=1 - 0.95 ** step)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    mo.adamw_step(pr[:n_decay], gr[:n_decay] * 0.5, mr[:n_decay], vr[:n_decay], step, lr, wd)
    mo.adamw_step(pr[n_decay:], gr[n_decay:] * 0.5, mr[n_decay:], vr[n_decay:], step, lr, 0.0)
    assert float((pd.cpu() - pr).abs().max()) < 2e-7 and float((md.cpu() - mr).abs().max()) < 1e-8
    assert float((vd.cpu() - vr).abs().max()) < 1e-9
    assert torch.equal(plp.cpu(), pd.cpu().to(dtype)) and bool((gd == 0).all())


# ------------------------------------------------------------------------------------ similarity search
def _run_topk(ops, q, x, w, k):
    Q, D = q.shape
    N = x.shape[0]
    qd, xd = dev(torch.from_numpy(q)), dev(torch.from_numpy(x))
    wd = dev(torch.from_numpy(w)) if w is not None else None
    tw, qn, xn = torch.empty(Q, D, device=DEV), torch.empty(Q, device=DEV), torch.empty(N, device=DEV)
    ops.weighted_norms(qd, wd, qn, tw)
    ops.weighted_norms(xd, wd, xn)
    nch = ops.cosine_topk_chunks(N, Q, D, k)
    ps = torch.empty(Q, nch, k, device=DEV)
    pi = torch.empty(Q, nch, k, device=DEV, dtype=torch.int64)
    ops.cosine_topk(tw, qn, xd, xn, k, 1e-6, 0, nch, ps, pi)
    os_, oi = torch.empty(Q, k, device=DEV), torch.empty(Q, k, device=DEV, dtype=torch.int64)
    ops.topk_merge(ps, pi, Q, nch, k, os_, oi)
    sc = torch.empty(Q, N, device=DEV)
    ops.cosine_scores(tw, qn, xd, xn, 1e-6, sc)
    torch.cuda.synchronize()
    return os_.cpu().numpy(), oi.cpu().numpy(), sc.cpu().numpy()


@pytest.mark.parametrize("Q,N,D,k,weighted", [(1, 5000, 768, 100, True), (5, 3001, 96, 10, False), (16, 4097, 768, 300, True),
                                              (70, 9000, 768, 100, True), (130, 2500, 64, 17, False), (3, 50, 32, 64, True)])
def test_cosine_topk_bit_exact(ops, Q, N, D, k, weighted):
    rng = np.random.default_rng(Q * 1000 + N)
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[7] = x[3]            # exact ties across rows
    x[N - 1] = x[3]
    if N > 2000:
        x[1999] = x[3]
    w = None
    if weighted:
        w = rng.random(D, dtype=np.float32) + 0.1
        w /= w.sum()
    ref_s, ref_i = so.cosine_topk_np(q, x, k, w)
    ref_sc = so.cosine_scores_np(q, x, w)
    s, i, sc = _run_topk(ops, q, x, w, k)
    assert np.array_equal(sc, ref_sc), "scores must be bit-identical to the fixed-order oracle"
    assert np.array_equal(i, ref_i)
    assert np.array_equal(s, ref_s)


def test_standardise_exact(ops):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1000, 96), dtype=np.float32) * 3 + 1
    mu, sd = x.mean(0), x.std(0, ddof=1)
    out = torch.empty(1000, 96, device=DEV)
    ops.standardise(dev(torch.from_numpy(x)), dev(torch.from_numpy(mu)), dev(torch.from_numpy(sd)), out)
    assert np.array_equal(out.cpu().numpy(), so.standardise_np(x, mu, sd))


def test_errors_are_reported(ops):
    a = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)  # K=12 is not a multiple of 8
    with pytest.raises(Exception) as e:
        ops.gemm(a, a, M=8, N=8, K=12, out_f32=torch.zeros(8, 8, device=DEV))
    assert "multiples" in str(e.value)


@pytest.mark.parametrize("split", [0, 2, 5])
def test_gemm_split_k_wgrad(ops, split):
    """wgrad-shaped launch (contraction over tokens) through the deterministic split-K path + fused bias gradient."""
    tokens, N, K = 1280, 136, 72          # dW[N,K] = dy[tokens,N]^T x[tokens,K]
    g = torch.Generator().manual_seed(split)
    dy, x = torch.randn(tokens, N, generator=g), torch.randn(tokens, K, generator=g)
    dyr, xr = dy.bfloat16().float(), x.bfloat16().float()
    ws = torch.zeros(4 * 1024 * 1024, device=DEV)
    outs = []
    for _ in range(2):
        dw = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        ops.gemm(dev(dy, torch.bfloat16), dev(x, torch.bfloat16), M=N, N=K, K=tokens, a_layout=ops.RC, b_layout=ops.RC,
                 lda=N, ldb=K, out_f32=dw, colsum_a=db, ws=ws, split_k=split)
        outs.append((dw.cpu(), db.cpu()))
    ref = dyr.double().T @ xr.double()
    scale = float((dyr.abs().double().T @ xr.abs().double()).max())
    assert float((outs[0][0].double() - ref).abs().max()) <= 2e-6 * scale
    assert float((outs[0][1].double() - dyr.double().sum(0)).abs().max()) <= 1e-5 * float(dyr.abs().sum(0).max())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # run-to-run deterministic


@pytest.mark.parametrize("b_layout", [0, 1])
def test_gemm_row_tail_launch_at_vit_l_token_counts(ops, b_layout, monkeypatch):
    """8320 token rows (mim_19: 128 x 65) x 1024 columns = 520 tiles of 128x128, one round of 512 plus 8: the dispatcher
    runs the last 128 rows as a second, split-K launch.  Forward-shaped (bias + fp32 residual) and data-gradient-shaped
    (row-contiguous weights, dGELU) launches against fp64 matmuls; the residual / aux / output row offsets of the tail
    launch are what this checks."""
    M, N, K = 8320, 1024, 2048
    g = torch.Generator().manual_seed(5 + b_layout)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    Ar, Br = A.bfloat16().float(), Bm.bfloat16().float()
    ws = torch.zeros(8 * 1024 * 1024, device=DEV)
    Ad = dev(A, torch.bfloat16)
    Bd = dev(Bm.T.contiguous() if not b_layout else Bm, torch.bfloat16)
    ref = (Ar.double() @ Br.double().T)
    scale = float(ref.abs().max())
    if b_layout:      # forward: bias + residual -> fp32
        resid = torch.randn(M, N, generator=g)
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(Ad, Bd, M=M, N=N, K=K, bias=dev(bias), resid=dev(resid), ldr=N, out_f32=out, ws=ws)
        want = ref + bias.double() + resid.double()
        assert float((out.cpu().double() - want).abs().max()) <= 1e-5 * scale
    else:             # data gradient: dX = dY . W (W row-contiguous) times dGELU(aux) -> bf16
        aux = torch.randn(M, N, generator=g)
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(Ad, Bd, M=M, N=N, K=K, a_layout=ops.KC, b_layout=ops.RC, lda=K, ldb=N, act=ops.ACT_DGELU, aux=dev(aux, torch.bfloat16),
                 ldaux=N, out=out, ws=ws)
        x = aux.bfloat16().double()
        dgelu = 0.5 * (1.0 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
        want = ref * dgelu
        assert float((out.float().cpu().double() - want).abs().max()) <= 1e-2 * scale
    torch.cuda.synchronize()


@pytest.mark.parametrize("b_l", [0, 1])
@pytest.mark.parametrize("shape", [(1000, 512, 256), (520, 768, 128), (256, 256, 1024)])
def test_gemm_256x256_eight_wave_tile(ops, b_l, shape):
    """csrc/gemm_pipe256.h (tile code 256256): persistent workgroups, ten-slot half-tile ring, two wave groups half a phase apart.
    Ragged row / column edges, the shortest k-loop it takes (two k-tiles: prologue = whole problem) and a longer one, k-contiguous
    and row-contiguous B, every epilogue it carries (bias, fp32 residual, GELU with pre-activation, dGELU, alpha, fp32 + bf16
    outputs) -- against fp64."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + K + b_l)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    T = torch.bfloat16
    Ar, Br = A.to(T).double(), B.to(T).double()
    Ad, Bd = dev(A, T), dev(B.T.contiguous() if b_l else B, T)
    kw = dict(M=M, N=N, K=K, a_layout=0, b_layout=b_l, tile=256256)
    prod = Ar @ Br.T
    scale = float((Ar.abs() @ Br.abs().T).max())
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Bd, out_f32=out, **kw)
    assert float((out.cpu().double() - prod).abs().max()) <= 2e-6 * scale
    out2 = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Bd, out_f32=out2, **kw)
    assert torch.equal(out, out2)                                      # run-to-run deterministic
    out32, outlp = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, alpha=0.5, bias=dev(bias), resid=dev(resid), ldr=N, out_f32=out32, out=outlp, **kw)
    ref = prod * 0.5 + bias.double() + resid.double()
    assert float((out32.cpu().double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((outlp.float().cpu().double() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    act, pre = torch.zeros(M, N, device=DEV, dtype=T), torch.zeros(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, bias=dev(bias), act=ops.ACT_GELU, out=act, out2=pre, **kw)
    v = (prod + bias.double()).float()
    assert float((pre.float().cpu() - v).abs().max()) <= 1e-2 * float(v.abs().max())
    assert float((act.float().cpu() - torch.nn.functional.gelu(v)).abs().max()) <= 1e-2 * float(v.abs().max())
    aux = torch.randn(M, N, generator=g)
    auxr = aux.to(T).float().requires_grad_(True)
    torch.nn.functional.gelu(auxr).sum().backward()
    dg = torch.zeros(M, N, device=DEV, dtype=T)
    ops.gemm(Ad, Bd, aux=dev(aux, T), ldaux=N, act=ops.ACT_DGELU, out=dg, **kw)
    refd = prod.float() * auxr.grad
    assert float((dg.float().cpu() - refd).abs().max()) <= 1e-2 * float(refd.abs().max())
    # outside its subset (a row-contiguous A that is not a weight gradient of whole tiles, a single k-tile) the explicit tile is
    # refused, not silently served by another kernel
    if b_l == 1 and M % 256 == 0 and N % 256 == 0:
        out.fill_(float("nan"))
        ops.gemm(dev(A.T.contiguous(), T), Bd, M=M, N=N, K=K, a_layout=1, b_layout=b_l, out_f32=out, tile=256256)
        assert float((out.cpu().double() - prod).abs().max()) <= 2e-6 * scale
    else:
        with pytest.raises(Exception):
            ops.gemm(dev(A.T.contiguous(), T), Bd, M=M, N=N, K=K, a_layout=1, b_layout=b_l, out_f32=out, tile=256256)
    with pytest.raises(Exception):
        ops.gemm(Ad[:, :64].contiguous(), dev(B[:, :64].contiguous(), T), M=M, N=N, K=64, out_f32=out, tile=256256)


@pytest.mark.parametrize("b_layout", [0, 1])
def test_gemm_256x256_chosen_at_vit_l_size_with_its_row_tail(ops, b_layout):
    """[8320 x 4096 x 1024] (mim_19's fc1 forward / fc2 data gradient) with the tile choice left open: 33 x 16 = 528 tiles of
    256 x 256 = two rounds of 256 + one row block of 16, which the dispatcher cuts off as a second launch (no workspace given:
    the one-launch two-k-group tail).  GELU with both outputs / dGELU, against the 128 x 128 ring kernel on the same inputs
    (different summation order: compared with a bf16-sized tolerance) and spot rows against fp64."""
    M, N, K = 8320, 4096, 1024
    g = torch.Generator(device=DEV).manual_seed(11 + b_layout)
    A = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    Bd = W if b_layout == 0 else W.T.contiguous()
    bias = torch.randn(N, device=DEV, generator=g)
    aux = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    outs = []
    for tile in (0, 128128):
        o1, o2 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16), torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        if b_layout == 0:
            ops.gemm(A, Bd, M=M, N=N, K=K, bias=bias, act=ops.ACT_GELU, out=o1, out2=o2, tile=tile)
        else:
            ops.gemm(A, Bd, M=M, N=N, K=K, a_layout=ops.KC, b_layout=ops.RC, lda=K, ldb=N, act=ops.ACT_DGELU, aux=aux, ldaux=N, out=o1, tile=tile)
        outs.append((o1.float(), o2.float()))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) <= 2e-2 * max(float(b.abs().max()), 1e-6)
    rows = [0, 255, 256, 4111, 8191, 8192, 8319]                       # both sides of the main / tail seam and of tile seams
    ref = A[rows].double() @ W.double().T
    if b_layout == 0:
        v = (ref + bias.double()).float()
        assert float((outs[0][1][rows] - v).abs().max()) <= 1e-2 * float(v.abs().max())
        assert float((outs[0][0][rows] - torch.nn.functional.gelu(v)).abs().max()) <= 1e-2 * float(v.abs().max())
    else:
        x = aux[rows].double()
        dgelu = 0.5 * (1.0 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
        want = (ref * dgelu).float()
        assert float((outs[0][0][rows] - want).abs().max()) <= 1e-2 * float(want.abs().max())


def test_simmim_mask_counts_match_reference_mask_generator():
    """The device MaskGenerator against masks the reference's own class drew (tests/golden/maskgen.npz, made by
    tests/golden/make_golden.py maskgen): fed the reference's ratio draw, the kernel masks exactly as many patches per
    channel as the reference did, for every draw of every geometry (incl. mim_19's 128/16/0.6)."""
    from sky_embeddings_amd import ops
    from tests.helpers import GOLDEN
    import os
    z = np.load(os.path.join(str(GOLDEN), "maskgen.npz"))
    for key in sorted({k.rsplit("/", 1)[0] for k in z.files}):
        size, p, C, mx = key.split("/")[1].split("_")
        size, p, C, mx = int(size), int(p), int(C), float(mx)
        u, masks = torch.from_numpy(z[key + "/u"]), torch.from_numpy(z[key + "/masks"]).float()
        grid = size // p
        noise = torch.rand(len(u), C, grid * grid, generator=torch.Generator().manual_seed(2)).cuda()
        out = torch.empty(len(u), C, size, size, device="cuda")
        ops.simmim_mask_from_noise(noise, u.cuda(), mx, grid, p, out)
        assert torch.equal(out[:, :, ::p, ::p].sum(dim=(2, 3)).cpu(), masks[:, :, ::p, ::p].sum(dim=(2, 3)))


@pytest.mark.parametrize("L,p,C,max_ratio", [(64, 8, 5, 0.9), (16, 16, 9, 0.6), (64, 16, 5, 0.6), (256, 4, 2, 1.0)])
def test_simmim_mask_from_noise_matches_oracle(L, p, C, max_ratio):
    """Device MaskGenerator (utils/dataloaders.py:197-219): per-sample ratio, ceil(L * ratio) patches per channel, an
    independent uniformly random subset per channel, expanded to pixels -- bit-equal to the oracle on the same draws,
    including tied noise values and the ratio extremes."""
    from oracle import mae_oracle as mo
    from sky_embeddings_amd import ops
    B = 7
    g = torch.Generator().manual_seed(L + C)
    noise = torch.rand(B, C, L, generator=g)
    noise[0, 0, :] = 0.5                                   # all tied: the first `count` patches are masked
    noise[1, 1, 3] = noise[1, 1, 9]
    u = torch.rand(B, generator=g)
    u[2], u[3] = 0.0, 0.999999
    ref = mo.simmim_mask_from_noise(noise, u, max_ratio, p)
    grid = int(round(L ** 0.5))
    out = torch.empty(B, C, grid * p, grid * p, device="cuda")
    ops.simmim_mask_from_noise(noise.cuda(), u.cuda(), max_ratio, grid, p, out)
    assert torch.equal(out.cpu(), ref)
    per_channel = out[:, :, ::p, ::p].sum(dim=(2, 3)).cpu()
    want = torch.tensor([int(np.ceil(np.float32(L * (float(u[b]) * max_ratio)))) for b in range(B)], dtype=torch.float32)
    assert torch.equal(per_channel, want[:, None].expand(B, C))


def test_augmentation_kernel_matches_oracle():
    """Device augmentation pipeline (csrc/augment.hip) vs the torch restatement of torchvision's tensor ops on the same drawn
    parameters: flips, resized crops of every kind the sampler produces (+ the identity crop and a thin one), brightness,
    noise, NaN channels, NaN pixels in the source (they spread to the pixels that interpolate from them)."""
    from oracle import augment_oracle as ao
    from sky_embeddings_amd.augment import Augmenter
    B, C, S, A = 5, 5, 64, 16
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(B, C, S, S, generator=g)
    imgs[1, 2] = float("nan")
    imgs[3, 0, 10:13, 40:47] = float("nan")
    aug = Augmenter(img_size=S, seed=11)
    params, nan_mask = aug.draw(B * (1 + A), C, S)
    params[3] = torch.tensor([1, 0, 0, 0, S, S, 1.0, 0.0])          # pure horizontal flip
    params[4] = torch.tensor([0, 1, 5, 9, 51, 55, 1.25, 0.01])
    params[5] = torch.tensor([1, 1, 0, 12, 64, 52, 0.8, 0.0])
    nan_mask[3] = 0
    noise = torch.randn(B * (1 + A), C, S, S, generator=g)
    ref = ao.augment(imgs, params, nan_mask, noise, A)
    out = aug.batch(imgs, A, params=params, nan_mask=nan_mask, noise=noise).cpu()
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert torch.equal(out[0], imgs[0]) and torch.equal(out[3], imgs[0].flip(-1))
    err = (torch.nan_to_num(out) - torch.nan_to_num(ref)).abs().max()
    assert float(err) <= 5e-6 * float(torch.nan_to_num(ref).abs().max()), float(err)
    # the drawn parameters respect torchvision's ranges
    h, w = params[:, 4], params[:, 5]
    assert bool(((h >= 1) & (h <= S) & (w >= 1) & (w <= S)).all()) and bool((params[:, 2] + h <= S).all()) and bool((params[:, 3] + w <= S).all())
    keep = torch.ones(len(params), dtype=torch.bool)
    keep[[3, 4, 5]] = False
    area = (h * w)[keep] / (S * S)
    assert float(area.min()) > 0.75 and float(area.max()) <= 1.0
    assert bool(((params[keep, 6] >= 0.8) & (params[keep, 6] <= 1.25)).all()) and bool(((params[keep, 7] >= 0) & (params[keep, 7] <= 0.01)).all())
    nbits = torch.tensor([bin(int(m)).count("1") for m in nan_mask])
    assert int(nbits.max()) <= 2 and set(nbits.tolist()) == {0, 1, 2}
    # drop-in call on one sample: an augmented copy of the same shape
    one = aug(imgs[0])
    assert one.shape == imgs[0].shape and not torch.equal(torch.nan_to_num(one), imgs[0])


@pytest.mark.parametrize("S,k", [(100, 100), (2048, 7), (25600, 100), (32768, 1), (40000, 100)])
def test_kth_largest_floor_is_the_kth_value_minus_one_ulp(ops, S, k):
    """Both selection kernels (register bisection for S <= 32 K, radix select above): out[q] = the k-th largest of row q, one
    ulp lower; NaNs rank as -inf, duplicates and infinities are ordinary values."""
    g = torch.Generator().manual_seed(S + k)
    x = torch.randn(5, S, generator=g) * 0.04
    x[1, :50] = x[1, 50:100]                      # duplicates around the top
    x[2, 3] = float("nan")
    x[2, 9] = float("inf")
    x[3] = x[3].abs() * -1.0                      # all negative
    x[4, : S // 2] = 0.125                        # one value repeated: the k-th may sit inside the run
    out = torch.empty(5, device=DEV)
    ops.kth_largest_floor(dev(x), k, out)
    ref = torch.where(torch.isnan(x), torch.full_like(x, float("-inf")), x).sort(dim=1, descending=True).values[:, k - 1]
    want = torch.nextafter(ref, torch.full_like(ref, float("-inf")))
    assert torch.equal(out.cpu(), want), (out.cpu().view(torch.int32), want.view(torch.int32), ref.view(torch.int32))
