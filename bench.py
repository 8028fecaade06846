#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started WITHOUT a
    launcher, `--gpus N` spawns the N ranks itself -- before this process makes any GPU call -- and fails if the node
    shows fewer than N devices)

Headline (BASELINE.json metric, configs[1]): MAE pretraining images/sec -- ViT-B/16 MAE, 5x64x64
cutouts, mask 0.75, batch 256 per GPU, bf16 MFMA GEMMs with fp32 accumulation / statistics /
master weights, synthetic N(0,1) cutouts clipped at -3 (seed 1234), reference init distribution.
One "step" = noise draw + forward + backward + (gradient all-reduce) + fused AdamW + cosine LR on
one resident minibatch.  Secondary: cosine top-k queries/sec over a 1M x 768 fp32 bank
(``search`` object).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md, spec)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}   # (f16 and bf16 share the dense MFMA peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE: 256)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP graph replay")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"],
                    help="GEMM operand format of the timed step: f16 (default: the 16-bit mode that holds loss / pixels within the reference "
                         "tolerance of 1e-3), bf16 (same kernels and rate, 6e-3 on the pixels) or f32 (exact fp32 MFMA chains)")
    ap.add_argument("--skip-search", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-feeder", action="store_true")
    ap.add_argument("--skip-f32", action="store_true", help="skip the timing of the fp32 parity mode (extra.f32_mode)")
    ap.add_argument("--skip-mim19", action="store_true", help="skip the BASELINE configs[4] leg (SimMIM ViT-L/16, 5x128x128)")
    ap.add_argument("--skip-leg", action="store_true", help=argparse.SUPPRESS)   # internal: the GEMM-skip measurement (run_skip_leg)
    ap.add_argument("--probe", action="store_true", help="also time every GEMM shape of the step alone (HIP-graph probe)")
    ap.add_argument("--bank-rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--topk", type=int, default=100)
    return ap.parse_args()


def ev_time_ms(fn, iters, stream=None):
    """Average duration of fn() in ms measured with HIP events on torch's current stream (the
    stream every libskyemb kernel is launched on)."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def graph_time_ms(fn, reps=20, replays=5):
    """Average duration of fn() in ms with `reps` calls captured into one HIP graph (the way the step runs them: no host
    launch floor between kernels), replayed `replays` times between two HIP events on the launch stream."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    return ev_time_ms(g.replay, replays) / reps


def gemm_step_probe(B, dtype, dev, iters=20):
    """Launch-level roofline of the dominant kernel: every MFMA GEMM launch of one step (forward KC.KC, dgrad KC.RC,
    wgrad RC.RC: one grouped launch per transformer block + three single ones), each timed with HIP events around a
    HIP-graph replay.
    achieved = sum(2*M*N*K * count) / sum(avg launch time * count); avg_launch_us per kind is what the
    rocprofv3 kernel_stats average of gemm_pipe_kernel<64,64,A_KC,B_KC,3,4> must agree with."""
    from sky_embeddings_amd import ops
    Me, Md = B * 5, B * 17
    layers = []
    for M, D, depth in ((Me, 768, 12), (Md, 512, 8)):
        layers += [(M, 3 * D, D, depth, "bias"), (M, D, D, depth, "resid"), (M, 4 * D, D, depth, "gelu"), (M, D, 4 * D, depth, "dgelu")]
    layers += [(B * 4, 768, 1280, 1, "bias"), (Me, 512, 768, 1, "bias"), (Md, 1280, 512, 1, "bias")]
    ws = torch.zeros(8 * 1024 * 1024, device=dev)
    kinds = {k: dict(launches=0, us=0.0, flop=0.0) for k in ("fwd_KC.KC", "dgrad_KC.RC", "wgrad_RC.RC")}
    keep = []                                      # grouped launches hold raw pointers: keep the operands alive
    groups = {}                                    # (M tokens, depth) -> gemm_args of that block's four weight gradients
    for M, N, K, cnt, epi in layers:
        x = torch.randn(M, K, device=dev).to(dtype)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dtype)
        dy = torch.randn(M, N, device=dev).to(dtype)
        bias = torch.zeros(N, device=dev)
        y, y2 = torch.empty(M, N, device=dev, dtype=dtype), torch.empty(M, N, device=dev, dtype=dtype)
        y32, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
        dx, aux = torch.empty(M, K, device=dev, dtype=dtype), torch.randn(M, K, device=dev).to(dtype)
        dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        keep.append((x, dy, dw, db))
        if epi == "gelu":
            fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ops.ACT_GELU, out=y, out2=y2)
        elif epi in ("resid", "dgelu"):
            fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, ws=ws)
        else:
            fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y)
        if epi == "dgelu":
            dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=ops.KC, b_layout=ops.RC, lda=N, ldb=K, act=ops.ACT_DGELU,
                                     aux=aux, ldaux=K, out=dx, ws=ws)
        else:
            dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=ops.KC, b_layout=ops.RC, lda=N, ldb=K, out=dx, ws=ws)
        wkw = dict(M=N, N=K, K=M, a_layout=ops.RC, b_layout=ops.RC, lda=N, ldb=K, out_f32=dw, colsum_a=db)
        timed = [("fwd_KC.KC", fwd), ("dgrad_KC.RC", dgrad)]
        if cnt > 1 and dtype != torch.float32:    # a transformer-block layer: its dW/db ride in the block's grouped launch
            groups.setdefault((M, cnt), []).append(ops.gemm_args(dy, x, **wkw))
        else:
            timed.append(("wgrad_RC.RC", lambda: ops.gemm(dy, x, ws=ws, **wkw)))
            kinds["wgrad_RC.RC"]["flop"] += 2.0 * M * N * K * cnt
        for kind, f in timed:
            ms = graph_time_ms(f, iters)
            kinds[kind]["launches"] += cnt
            kinds[kind]["us"] += 1e3 * ms * cnt
            if kind != "wgrad_RC.RC":
                kinds[kind]["flop"] += 2.0 * M * N * K * cnt
    for (M, cnt), args in groups.items():
        grp = ops.GemmGroup(args, dev)
        ms = graph_time_ms(grp.launch, iters)
        kinds["wgrad_RC.RC"]["launches"] += cnt
        kinds["wgrad_RC.RC"]["us"] += 1e3 * ms * cnt
        kinds["wgrad_RC.RC"]["flop"] += sum(2.0 * a.M * a.N * a.K for a in args) * cnt
    tot_us = sum(v["us"] for v in kinds.values())
    tot_fl = sum(v["flop"] for v in kinds.values())
    per_kind = {k: dict(launches_per_step=v["launches"], avg_launch_us=v["us"] / v["launches"], tflops=v["flop"] / v["us"] / 1e6)
                for k, v in kinds.items()}
    return dict(kernel="gemm_pipe_kernel<64|128x64, ...> / gemm_pipe_group_kernel (the four dW of a block per launch) (+ splitk_reduce_kernel)", launches_per_step=sum(v["launches"] for v in kinds.values()),
                ms_per_step=tot_us / 1e3, flop_per_step=tot_fl, tflops=tot_fl / tot_us / 1e6, kinds=per_kind)


def gemm_flops_per_step(cfg, B, mask_ratio=0.75):
    """Algorithmic 2*M*N*K of every MFMA GEMM launch of one step (forward, data gradient, weight gradient of every
    Linear, the patch embedding on the kept patches only): the `executed` figure of MAEEngine.flops_per_image minus the
    attention core."""
    L, D, Dd, pv = cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim
    keep = int(L * (1 - mask_ratio))
    Ne, Nd, r = 1 + keep, 1 + L, cfg.mlp_ratio
    lin = lambda n, d, depth: depth * 2 * n * d * (3 * d + d + 2 * r * d)
    fwd = lin(Ne, D, cfg.depth) + 2 * Ne * D * Dd + lin(Nd, Dd, cfg.decoder_depth) + 2 * Nd * Dd * pv + 2 * keep * pv * D
    return 3.0 * fwd * B


MEASURE_SO = os.path.join(ROOT, "sky_embeddings_amd", "libskyemb_measure.so")


def run_skip_leg(args):
    """The GEMM family's in-step time by difference: the step with and without its MFMA GEMM launches (skyemb_debug_skip).  That
    switch is compiled into libskyemb_measure.so only, so the leg runs in a child process that loads that library (SKYEMB_LIB); both
    halves of each difference come from that one process.  Fails loudly when the measurement library has not been built."""
    import subprocess
    if not os.path.exists(MEASURE_SO):
        raise RuntimeError(f"{MEASURE_SO} is missing (make -C sky_embeddings_amd/csrc builds it beside libskyemb.so)")
    env = dict(os.environ, SKYEMB_LIB=MEASURE_SO)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--skip-leg", "--batch", str(args.batch), "--dtype", args.dtype, "--steps", str(max(args.steps, 20))]
    if args.no_graph:
        cmd.append("--no-graph")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if res.returncode != 0:
        raise RuntimeError("bench.py --skip-leg failed:\n" + res.stderr[-2000:])
    return json.loads(res.stdout.strip().splitlines()[-1])


def skip_leg(args):
    """Child of run_skip_leg (measurement library loaded): config A step, fused-optimiser schedule and separate-AdamW schedule, each
    with and without the GEMM launches; HIP events over `steps` steps.  Prints one JSON object."""
    from sky_embeddings_amd import _lib
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    assert os.path.realpath(_lib.SO_PATH) == os.path.realpath(MEASURE_SO), _lib.SO_PATH
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
    eng = MAEEngine(cfg, device=dev, compute_dtype=DTYPES[args.dtype], seed=0)
    torch.manual_seed(1234)
    opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    sched = CosineLR(opt, 1_000_000, eta_min=1e-4 / 1e7)
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234)
    pool = [torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0).to(dev) for _ in range(2)]

    def time_step(fused, skip):
        # (a new TrainStep per leg: the switch acts when the launches are CAPTURED, a captured graph replays what it recorded)
        assert _lib.lib().skyemb_debug_skip(1 if skip else 0) >= 0
        try:
            s_ = TrainStep(eng, opt, sched, B, mask_ratio=0.75, use_graph=not args.no_graph, world_size=1, fused_adamw=fused)
            for i in range(5):
                s_(pool[i % 2])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.steps):
                s_(pool[i % 2])
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / args.steps, s_.fused_adamw
        finally:
            _lib.lib().skyemb_debug_skip(0)

    out = {"library": "libskyemb_measure.so", "steps": args.steps}
    for name, fused in (("fused", True), ("separate", False)):
        with_ms, was_fused = time_step(fused, False)
        bare_ms, _ = time_step(fused, True)
        out[name] = dict(with_gemm_ms=with_ms, without_gemm_ms=bare_ms) if (was_fused == fused) else None
    print(json.dumps(out))


def bench_pretrain(args, rank, world, dev):
    from sky_embeddings_amd import _lib
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    dtype = DTYPES[args.dtype]
    cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
    eng = MAEEngine(cfg, device=dev, compute_dtype=dtype, seed=0)       # same weights on every rank
    torch.manual_seed(1234 + rank)                                      # ... but its own masking-noise stream (seed = base + rank)
    opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    sched = CosineLR(opt, 1_000_000, eta_min=1e-4 / 1e7)
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    pool = [torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0).to(dev) for _ in range(2)]
    step = TrainStep(eng, opt, sched, B, mask_ratio=0.75, use_graph=not args.no_graph, world_size=world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, n):
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            out = fn(pool[i % 2])
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        return dt, e0.elapsed_time(e1) / n, out

    for i in range(args.warmup):
        step(pool[i % 2])
    dt, gpu_ms, loss = timed(step, args.steps)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    executed, algorithmic = eng.flops_per_image(0.75)
    out = dict(fused_adamw=step.fused_adamw, adamw_side=getattr(step, "adamw_side", None), ms_per_step=1e3 * dt / args.steps, images_per_sec=world * B * args.steps / dt, gpu_ms_per_step=gpu_ms,
               loss=float(loss), flops_per_image_executed=executed, flops_per_image_reference=algorithmic, B=B)
    if rank == 0:
        # In-step time of the dominant kernel family: the same step with the MFMA GEMM launches left out (the C ABI's
        # measurement switch), HIP events over the same number of steps; the difference is what the GEMM launches take
        # INSIDE the step (cold weights, launch gaps and all) -- the figure rocprofv3's kernel_stats must agree with.
        skip = run_skip_leg(args)          # (a process of its own on libskyemb_measure.so: the product library has no such switch)
        out["skip_leg"] = skip
        bare_ms = skip["fused"]["without_gemm_ms"] if skip["fused"] else skip["separate"]["without_gemm_ms"]
        with_ms = skip["fused"]["with_gemm_ms"] if skip["fused"] else skip["separate"]["with_gemm_ms"]
        gf = gemm_flops_per_step(cfg, B)
        gemm_ms = with_ms - bare_ms        # both legs on the measurement build, same process, same box state
        # forward, dgrad singles; 20 grouped weight-gradient launches + the single ones (patch_embed; decoder_pred / decoder_embed unless
        # they ride in a grouped launch as further problems: engine._extra_wgrad_layers)
        folded = max((len(w_.get("folded_wgrads", ())) for k_, w_ in eng._ws.items() if k_[-1] is True), default=0)
        launches = 83 * 2 + 20 + 3 - folded
        out["gemm_in_step"] = dict(ms_per_step=gemm_ms, step_without_gemm_ms=bare_ms, flop_per_step=gf, launches_per_step=launches,
                                   avg_launch_us=1e3 * gemm_ms / launches, tflops=gf / gemm_ms / 1e9)
        if args.probe:
            out["gemm_probe"] = gemm_step_probe(B, dtype, dev)
        out["phases"] = hbm_phases(eng, opt, step, B)
        if world == 1:
            # a longer timed region beside the driver's (>= 100 steps), HIP events on the launch stream
            n_long = max(100, args.steps)
            _, long_ms, _ = timed(step, n_long)
            out["long_run"] = dict(steps=n_long, ms_per_step=long_ms, images_per_sec=B / long_ms * 1e3)
            out["staged"] = staged_schedule_price(eng, opt, sched, B, pool, dev, args, skip)
            if B == 256 and not args.no_graph:
                out["batch_sweep"] = batch_sweep(eng, opt, sched, dev, rank, long_ms)
            if args.dtype in ("f16", "bf16") and not args.no_graph:
                out["operand_format_ab"] = operand_format_ab(cfg, args.dtype, B, pool, dev, step)
            if step.fused_adamw and not args.no_graph:
                out["optimizer_placement"] = placement_ab(lambda: MAEEngine(cfg, device=dev, compute_dtype=dtype, seed=0), B, pool, dev, step,
                                                          ["1", "0", "ln_separate=SKYEMB_LN_SIDE=0"] if os.environ.get("SKYEMB_BENCH_PLACEMENT_ALL")
                                                          else ["0", "ln_separate=SKYEMB_LN_SIDE=0", "no_fold=SKYEMB_FOLD_WGRADS=0", "no_prefetch=SKYEMB_PREFETCH=0",
                                                                "tiles_per_xcd=SKYEMB_GROUP_XCD_ORDER=1"] + os.environ.get("SKYEMB_BENCH_EXTRA_VARIANTS", "").split())
        if world == 1 and not args.skip_feeder:
            out["feeder"] = bench_feeder(args, dev, step, B)
    return out, eng


def operand_format_ab(cfg, timed, B, pool, dev, step_timed, rounds=3, n=30):
    """Interleaved A/B, one process: the SAME step with fp16 and with bf16 GEMM operands (csrc/lp_twin.h: one kernel source compiled
    for both; layouts, tiles, launch counts identical).  `step_timed` is the TrainStep the headline timed."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    other = "bf16" if timed == "f16" else "f16"
    eng2 = MAEEngine(cfg, device=dev, compute_dtype=DTYPES[other], seed=0)
    opt2 = FusedAdamW(eng2, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    s2 = TrainStep(eng2, opt2, CosineLR(opt2, 1_000_000, eta_min=1e-4 / 1e7), B, mask_ratio=0.75, use_graph=True, world_size=1)
    steps = {timed: step_timed, other: s2}
    for s_ in steps.values():
        for i in range(5):
            s_(pool[i % 2])
    res = {k: [] for k in steps}
    for _ in range(rounds):
        for name, s_ in steps.items():
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                s_(pool[i % 2])
            e1.record()
            e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / n)
    del steps, s2, opt2, eng2
    torch.cuda.empty_cache()
    mean = {k: sum(v) / len(v) for k, v in res.items()}
    return dict(ms_per_step=res, mean_ms=mean, images_per_sec={k: B / v * 1e3 for k, v in mean.items()},
                note="the step with fp16 / bf16 MFMA operands, interleaved rounds in one process (HIP events); parity.f16 / parity.bf16 are the "
                     "two formats' errors against the fp32 CPU oracle")


def placement_ab(make_engine, B, pool, dev, step_default, others, load=None, rounds=3, n=30):
    """Interleaved A/B, one process: where the AdamW step of the transformer blocks' weights runs.  `step_default` is the TrainStep the
    headline timed (policy "auto": side jobs where the carrying launch leaves compute units idle -- the 256 x 256 groups of ViT-L --
    and the launch's own epilogue elsewhere); `others` are policies built on further engines with the same weights: "0" = epilogue
    everywhere (the round 3-4 form), "1" = side jobs everywhere, "dec" = side jobs carried by the decoder's launches only.
    HIP events, `rounds` x `n` steps each, alternating."""
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    steps = {"auto": (step_default, None)}
    for pol in others:
        # ("name=ENV=value": the shipped policy with one library switch flipped while the variant's launches are planned)
        env = dict([pol.split("=", 2)[1:]]) if "=" in pol else {}
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            eng2 = make_engine()
            opt2 = FusedAdamW(eng2, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
            s2 = TrainStep(eng2, opt2, CosineLR(opt2, 1_000_000, eta_min=1e-4 / 1e7), B, mask_ratio=0.75, use_graph=True, world_size=1,
                           adamw_side=None if env else pol)
            if load is not None:
                load(s2)
            for i in range(2):                       # (plans are made when the first step builds the workspace)
                s2(pool[i % 2]) if pool is not None else s2()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        steps[pol.split("=")[0]] = (s2, (eng2, opt2))
    call = (lambda s, i: s(pool[i % 2])) if pool is not None else (lambda s, i: s())
    for name, (s, _) in steps.items():
        for i in range(5):
            call(s, i)
    res = {k: [] for k in steps}
    for _ in range(rounds):
        for name, (s, _) in steps.items():
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                call(s, i)
            e1.record()
            e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / n)
    side_launches = {}
    for name, (s, _) in steps.items():
        ws = [w for k, w in s.engine._ws.items() if k[-1] is True]
        side_launches[name] = int(ws[-1].get("adamw_side_launches", 0)) if ws else None
    steps.clear()
    torch.cuda.empty_cache()
    return dict(ms_per_step={k: v for k, v in res.items()}, mean_ms={k: sum(v) / len(v) for k, v in res.items()}, side_launches_per_step=side_launches,
                note="AdamW of the transformer blocks' weights: 'auto' (shipped) = side job of the NEXT block's grouped weight-gradient launch where "
                     "that launch leaves compute units idle (256 x 256 tiles), else the epilogue of the block's own launch; '0' = epilogue everywhere; "
                     "'1' = side jobs everywhere; 'dec' = carried by the decoder's launches only; 'tiles_per_problem' = the shipped policy with the 256 x 256 "
                     "groups' tiles laid out per problem instead of per XCD (SKYEMB_GROUP_XCD_ORDER=0); 'tiles_per_xcd' = the shipped policy with the RING-TILE "
                     "grouped launches' tiles laid out per XCD too (an eighth of the concatenated list per XCD: SKYEMB_GROUP_XCD_ORDER=1; built in round 6, not shipped); 'ln_separate' = the shipped policy with every block's norm1 "
                     "backward as its own launch behind the grouped weight-gradient launch instead of side workgroups inside it (SKYEMB_LN_SIDE=0); 'no_fold' = the "
                     "shipped policy with the single weight gradients outside the blocks (decoder_pred, decoder_embed / the SimMIM head) as their own "
                     "split-K launches instead of fifth problems of a grouped launch (SKYEMB_FOLD_WGRADS=0); 'no_prefetch' = the shipped policy without the "
                     "prefetch hints (every GEMM launch touching the next GEMM's weights: skyemb_gemm_args.prefetch; SKYEMB_PREFETCH=0).  "
                     "Interleaved rounds in one process")


def batch_sweep(eng, opt, sched, dev, rank, ms_256):
    """The same engine, weights and step at B = 512 and 1024 beside the headline's B = 256: step time against the batch is a straight
    line T(B) = F + V B -- F the per-launch fixed cost of ~340 dependent launches, V the kernels' throughput -- and the bench line
    shows both (extra.batch_sweep.fixed_ms / per_image_us).  Not the headline: BASELINE configs[1] prescribes bs = 256 per GPU."""
    from sky_embeddings_amd.train_step import TrainStep
    st = eng.store
    snap = [t.clone() for t in (st.p, st.m, st.v, st.p_lp)]
    counters = (opt.step_count, sched.last_epoch)
    res = {"256": dict(ms_per_step=ms_256, images_per_sec=256 / ms_256 * 1e3)}
    executed, _ = eng.flops_per_image(0.75)
    try:
        for Bs in (512, 1024):
            g = torch.Generator(device="cpu").manual_seed(4321 + rank)
            batch = torch.randn(Bs, 5, 64, 64, generator=g).clamp_(min=-3.0).to(dev)
            s = TrainStep(eng, opt, sched, Bs, mask_ratio=0.75, use_graph=True, world_size=1)
            for _ in range(5):
                s(batch)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                s(batch)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 30
            res[str(Bs)] = dict(ms_per_step=ms, images_per_sec=Bs / ms * 1e3, frac_of_bf16_peak=Bs / ms * executed / 1e9 / PEAK_BF16_TFLOPS)
            del s, batch
            for k in [k for k in eng._ws if k[0] == Bs]:
                del eng._ws[k]
            torch.cuda.empty_cache()
    finally:
        for dst, src in zip((st.p, st.m, st.v, st.p_lp), snap):
            dst.copy_(src)
        opt.step_count, sched.last_epoch = counters
        sched._apply()
    # least-squares line through the three points
    xs = [256.0, 512.0, 1024.0]
    ys = [res[str(int(x))]["ms_per_step"] for x in xs]
    mx, my = sum(xs) / 3, sum(ys) / 3
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    res["fixed_ms"], res["per_image_us"] = my - slope * mx, 1e3 * slope
    res["per_image_frac_of_bf16_peak"] = executed / (slope * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
    res["note"] = ("T(B) = fixed_ms + per_image_us * B through B = 256 / 512 / 1024 on the same engine (HIP graph, optimiser in the weight-"
                   "gradient epilogues); per_image_frac_of_bf16_peak = executed FLOPs per image / per_image_us: the kernels' own rate "
                   "once the per-launch fixed cost is paid")
    return res


def f32_mode_timing(dev, rank, B=256):
    """The mode that meets north_star's 1e-3 pixel tolerance (exact fp32 MFMA chains, csrc/gemm.hip), timed on the headline's
    workload: same model, batch and step, HIP graph; its AdamW is the separate launch."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
    eng = MAEEngine(cfg, device=dev, compute_dtype=torch.float32, seed=0)
    opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    step = TrainStep(eng, opt, CosineLR(opt, 1_000_000, eta_min=1e-4 / 1e7), B, mask_ratio=0.75, use_graph=True, world_size=1)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    batch = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0).to(dev)
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        loss = step(batch)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    executed, _ = eng.flops_per_image(0.75)
    res = dict(ms_per_step=ms, images_per_sec=B / ms * 1e3, loss=float(loss), tflops=B / ms * executed / 1e9,
               frac_of_f32_mfma_peak=B / ms * executed / 1e9 / PEAK_F32_MFMA_TFLOPS,
               note="compute_dtype = f32: v_mfma_f32_16x16x4_f32 GEMMs (exact fp32 fma chains), fp32 attention; parity.f32 is this mode's error")
    del step, opt, eng
    torch.cuda.empty_cache()
    return res


def staged_schedule_price(eng, opt, sched, B, pool, dev, args, skip):
    """What the N > 1 schedule costs in COMPUTE on one GPU (no 8-GPU node is guaranteed to the driver): the step as 8 stage
    graphs (decoder, six encoder groups, embedding) with the per-stage gradient casts into the bf16 communication mirror --
    everything the data-parallel step does except the collectives -- beside the monolithic graph; also with fp32 communication
    buffers (no casts).  Bytes per stage = what each stage's all-reduce would move."""
    from sky_embeddings_amd.train_step import TrainStep
    st = eng.store
    snap = [t.clone() for t in (st.p, st.m, st.v, st.p_lp)]
    counters = (opt.step_count, sched.last_epoch)
    grad_scale = opt.grad_scale
    res = {}
    try:
        # the monolithic step with the separate AdamW launch (what every rank of a data-parallel run executes per stage set)
        s1 = TrainStep(eng, opt, sched, B, mask_ratio=0.75, use_graph=not args.no_graph, world_size=1, fused_adamw=False)
        for i in range(5):
            s1(pool[i % 2])
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            s1(pool[i % 2])
        e1.record()
        e1.synchronize()
        sep_ms = e0.elapsed_time(e1) / 50
        # ... and the same step with the GEMM launches left out: the GEMM family's in-step time without optimiser work in it
        # (measured by run_skip_leg on the measurement build; handed in through `skip`)
        sep_bare = skip["separate"]["without_gemm_ms"]
        sep_with = skip["separate"]["with_gemm_ms"]
        res["monolithic_separate_adamw"] = dict(ms_per_step=sep_ms, step_without_gemm_ms=sep_bare, gemm_ms_per_step=sep_with - sep_bare,
                                                measurement_build_ms_per_step=sep_with)
        del s1
        for comm in ("bf16", "f32"):
            s2 = TrainStep(eng, opt, sched, B, mask_ratio=0.75, use_graph=not args.no_graph, world_size=1, staged=True,
                           n_encoder_groups=6, grad_comm=comm)
            for i in range(5):
                s2(pool[i % 2])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(50):
                s2(pool[i % 2])
            e1.record()
            e1.synchronize()
            elem = 2 if comm == "bf16" else 4
            res[comm] = dict(ms_per_step=e0.elapsed_time(e1) / 50, stages=len(s2.stages),
                             bytes_per_stage=[sum(e - s for s, e in ranges) * elem for _, ranges in s2.stages])
            del s2
        # the N > 1 default since round 6: the optimiser sharded over the ranks (reduce-scatter -> AdamW on this rank's eighth of
        # every block-weight range -> all-gather of the 16-bit shadow).  Priced here as rank 0 of 8, collectives left out: the
        # compute one rank of an 8-GPU job executes per step (the parameters of this leg are wrong by design and restored below)
        s3 = TrainStep(eng, opt, sched, B, mask_ratio=0.75, use_graph=not args.no_graph, world_size=1, staged=True, n_encoder_groups=6,
                       grad_comm="bf16", shard_optimizer=True, shard_emulate=(0, 8))
        assert s3.shard_optimizer
        for i in range(5):
            s3(pool[i % 2])
        torch.cuda.synchronize(dev)
        e0.record()
        for i in range(50):
            s3(pool[i % 2])
        e1.record()
        e1.synchronize()
        res["sharded_optimizer_rank0_of_8"] = dict(ms_per_step=e0.elapsed_time(e1) / 50,
                                                   optimizer_elements_per_rank=sum(max(c.numel(), 0) for c in s3._own.values()) +
                                                   sum(e - s for s, e in s3.stages[-1][1]), optimizer_elements_total=st.n)
        del s3
    finally:
        for dst, src in zip((st.p, st.m, st.v, st.p_lp), snap):
            dst.copy_(src)
        opt.step_count, sched.last_epoch = counters
        sched._apply()
        opt.grad_scale = grad_scale
    res["note"] = ("TrainStep(staged=True, n_encoder_groups=6) at world 1: stage graphs + skyemb_cast launches, no collectives; the "
                   "8-GPU step = sharded_optimizer_rank0_of_8 + the exposed part of the last stages' reduce-scatter / all-gather "
                   "(DESIGN.md section 2; bf16 / f32 = the replicated schedule of rounds 2-5: every rank steps every parameter)")
    return res


def bench_mim19(args, dev):
    """BASELINE configs[4]: configs/mim_19.ini -- SimMIM ViT-Large/16 on 5x128x128 cutouts, mask ratio 0.6, bs 128, bf16."""
    import configparser
    import numpy as np
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.simmim_engine import SimMIMEngine
    from sky_embeddings_amd.train_step import TrainStep
    ini = configparser.ConfigParser()
    ini.read(os.path.join(ROOT, "configs", "mim_19.ini"))
    a, t = ini["ARCHITECTURE"], ini["TRAINING"]
    cfg = config_for(a["model_type"], img_size=int(a["img_size"]), patch_size=int(a["patch_size"]), in_chans=int(a["num_channels"]),
                     embed_dim=int(a["embed_dim"]), norm_pix_loss=t.getboolean("norm_pix_loss"), loss_fn=t["loss_fn"])
    B = int(t["batch_size"])
    name = getattr(args, "dtype", None) or os.environ.get("SKYEMB_DTYPE", "f16")       # (args = None: tools/mim19_bench.py)
    lp = DTYPES[name] if name in ("f16", "bf16") else torch.bfloat16                   # (the headline's 16-bit operand format)
    eng = SimMIMEngine(cfg, device=dev, compute_dtype=lp, seed=0)
    opt = FusedAdamW(eng, lr=float(t["init_lr"]), betas=(0.9, 0.95), weight_decay=float(t["weight_decay"]))
    step = TrainStep(eng, opt, CosineLR(opt, 1_000_000), B)
    g = torch.Generator(device=dev).manual_seed(19)
    x = torch.randn(B, cfg.in_chans, cfg.img_size, cfg.img_size, device=dev, generator=g).clamp_(min=-3.0)
    L, p = cfg.num_patches, cfg.patch_size
    count = int(np.ceil(L * float(t["max_mask_ratio"])))
    order = torch.rand(B, cfg.in_chans, L, device=dev, generator=g).argsort(dim=2)
    m = (order < count).float().view(B, cfg.in_chans, cfg.grid, cfg.grid).repeat_interleave(p, 2).repeat_interleave(p, 3).contiguous()
    step.load_batch(x, m)                                  # the batch is resident in the step's input buffers when timing starts
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize(dev)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        loss = step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    executed, algorithmic = eng.flops_per_image()
    placement = None
    if step.fused_adamw and not os.environ.get("SKYEMB_BENCH_NO_AB"):       # (tools/mim19_bench.py under the profiler: the shipped step alone)
        placement = placement_ab(lambda: SimMIMEngine(cfg, device=dev, compute_dtype=lp, seed=0), B, None, dev, step,
                                 ["0", "ln_separate=SKYEMB_LN_SIDE=0", "no_fold=SKYEMB_FOLD_WGRADS=0", "no_prefetch=SKYEMB_PREFETCH=0"] + (["tiles_per_problem=SKYEMB_GROUP_XCD_ORDER=0"] if os.environ.get("SKYEMB_BENCH_PLACEMENT_ALL") else []),
                                 load=lambda s: s.load_batch(x, m), rounds=2, n=10)
    res = dict(workload="configs/mim_19.ini: SimMIM ViT-Large/16, 5x128x128, 39 of 64 patches masked per channel (ratio 0.6), "
                        f"bs={B}, L1 + norm-pix, AdamW+cosine, bf16", ms_per_step=ms, images_per_sec=B / ms * 1e3, optimizer_placement=placement,
               tflops=B / ms * executed / 1e9, frac_of_bf16_peak=B / ms * executed / 1e9 / PEAK_BF16_TFLOPS,
               flops_per_image_executed=executed, params=int(eng.store.n), loss=float(loss), **mim19_pmc_record())
    del step, opt, eng
    torch.cuda.empty_cache()
    return res


def hbm_phases(eng, opt, step, B):
    """Secondary HBM-bound phases of the step, each timed alone with HIP events: bytes are the algorithmic
    ones (AdamW: 28 B fp32 state + 2 B bf16 shadow per parameter; patch gather: kept patches in, bf16 rows out; loss:
    cutouts + predictions in, d pred out)."""
    from sky_embeddings_amd import ops
    cfg, st = eng.cfg, eng.store
    keep = int(cfg.num_patches * 0.25)
    w = eng._ws[(B, keep, True)]
    imgs = step.imgs
    res = {}
    t = ev_time_ms(opt.step, 10)
    opt.step_count -= 10                      # probe only: the moments move, the step counter is restored
    res["adamw"] = dict(ms=t, gbs=st.n * 30 / t / 1e6, bytes=st.n * 30)
    f = lambda: ops.patch_gather(imgs, st.param("patch_mask_values"), w["ids_keep"], w["patches"], cfg.patch_size, keep,
                                 cfg.pixel_mean, cfg.pixel_std)
    t = graph_time_ms(f)
    nb = B * keep * cfg.patch_dim * (4 + 2)
    res["patch_gather"] = dict(ms=t, gbs=nb / t / 1e6, bytes=nb)
    f = lambda: ops.masked_patch_loss(imgs, w["pred"], w["mask"], w["loss"], w["dpred"], None, eng.code, w["loss_ws"],
                                      cfg.patch_size, 1, cfg.pixel_mean, cfg.pixel_std, cfg.norm_pix_loss, cfg.loss_fn != "mse")
    t = graph_time_ms(f)
    nb = imgs.numel() * 4 + w["pred"].numel() * 4 + w["dpred"].numel() * 2
    res["masked_patch_loss"] = dict(ms=t, gbs=nb / t / 1e6, bytes=nb)
    return res


def bench_feeder(args, dev, step, B):
    """HDF5 -> HBM feeder (row a1): synthetic cutout file in the reference schema, the batched feeder alone and in the
    training loop (images/sec including file gather + H2D + clip; page cache warm)."""
    import tempfile
    from sky_embeddings_amd import hdf5_lite
    from sky_embeddings_amd.feeder import CutoutFeeder
    n = 8192
    with tempfile.TemporaryDirectory() as d:
        path = hdf5_lite.make_synthetic_cutouts(os.path.join(d, "cutouts.h5"), n=n, seed=1234)
        fd = CutoutFeeder(path, B, 64, dev, shuffle=True, seed=0, epochs=1000)
        it = iter(fd)
        for _ in range(8):
            next(it)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(64):
            next(it)
        torch.cuda.synchronize(dev)
        alone = 64 * B / (time.perf_counter() - t0)
        for _ in range(5):
            step(next(it)[0])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(next(it)[0])
        torch.cuda.synchronize(dev)
        fed = args.steps * B / (time.perf_counter() - t0)
        it.close()
    return dict(file_cutouts=n, feeder_alone_images_per_sec=alone, feeder_alone_gbs=alone * 5 * 64 * 64 * 4 / 1e9,
                train_images_per_sec_with_feeder=fed,
                note="mmap'd contiguous HDF5 (page cache) -> native threaded gather -> pinned ring -> one H2D per batch -> "
                     "device clip/crop, feeding the same TrainStep; `value` itself is measured with the batch resident in HBM")


def bench_search(args, rank, world, dev):
    from sky_embeddings_amd import ops
    from sky_embeddings_amd.search import PreparedBank, cosine_topk
    N, D, k = args.bank_rows, 768, args.topk
    rows = (N + world - 1) // world
    lo, hi = rank * rows, min(N, (rank + 1) * rows)
    g = torch.Generator(device=dev).manual_seed(2024)
    # every rank draws the same global stream and keeps its shard (bit-identical to the 1-GPU bank)
    bank = torch.empty(hi - lo, D, device=dev)
    chunk = 50_000
    pos = 0
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        blk = torch.randn(e - s, D, device=dev, generator=g)
        a, b = max(s, lo), min(e, hi)
        if a < b:
            bank[a - lo:b - lo] = blk[a - s:b - s]
    gq = torch.Generator(device=dev).manual_seed(2025)
    queries = torch.randn(args.queries, D, device=dev, generator=gq)
    gw = torch.Generator(device=dev).manual_seed(7)
    w = 1.0 / (torch.rand(D, device=dev, generator=gw) + 0.5) ** 2
    w = w / w.sum()
    pb = PreparedBank(bank, w, idx_offset=lo)
    res = {}
    # (20 back-to-back searches for the sub-millisecond cases: with 5, the host's start-up latency and the closing sync were
    # ~20 us of every 0.62 ms search)
    for label, Q, iters in (("q1", 1, 20), ("q_small", 16, 20), ("q_large", args.queries, 2)):
        q = queries[:Q]
        cosine_topk(q, pb, k, world_size=world)  # warm-up
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        st = {}
        for _ in range(iters):
            s, i = cosine_topk(q, pb, k, world_size=world, stats=st)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / iters
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        # dominant kernel alone, HIP events on the launch stream
        tw = torch.empty(Q, D, device=dev)
        qn = torch.empty(Q, device=dev)
        ops.weighted_norms(q.contiguous(), w, qn, tw)
        nch = ops.cosine_topk_chunks(hi - lo, Q, D, k)
        ps = torch.empty(Q, nch, k, device=dev)
        pi = torch.empty(Q, nch, k, device=dev, dtype=torch.int64)
        from sky_embeddings_amd.search import pruning_floor
        thr0 = pruning_floor(tw, qn, pb, k, 1e-6)
        kms = ev_time_ms(lambda: ops.cosine_topk(tw, qn, bank, pb.norms, k, 1e-6, lo, nch, ps, pi, thr0), iters)
        bank_bytes = (hi - lo) * D * 4
        # bank + row norms + queries + the [Q, k] result (the per-wave candidate lists are a few rows each since round 3: a list
        # ends at its first negative index, the padding behind it is no longer written or read)
        kbytes = bank_bytes + (hi - lo) * 4 + Q * D * 4 + Q * k * 12
        # `sec` is the path cosine_topk took (st["path"]); kernel_* time the EXACT kernel on the same inputs -- the whole
        # job for Q <= 16, and for many queries the single-stage path the two-stage one replaced (and falls back to)
        res[label] = dict(Q=Q, sec=dt, queries_per_sec=Q / dt, path=st.get("path"), redone=st.get("redone"),
                          effective_tflops=2.0 * Q * (hi - lo) * D / dt / 1e12,
                          kernel="cosine_topk_stream_kernel" if Q <= 16 else "cosine_topk_kernel<64,256,8> (exact, single stage)",
                          kernel_ms=kms, kernel_bytes=kbytes, kernel_hbm_gbs=kbytes / kms / 1e6,
                          kernel_tflops=2.0 * Q * (hi - lo) * D / kms / 1e9, checksum=int(i.sum().item() % (1 << 31)))
    return res, (queries, w, bank)


def pmc_traffic(args):
    """HBM-side bytes per Q = 16 bank-pass launch from the newest committed PMC summary (profiles/rNN_topk_stream_pmc.json), or None."""
    if args.bank_rows != 1_000_000 or args.topk != 100 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        return None                                        # (counters cannot be read from inside the timed process: the profiles/ figure of the same workload)
    rec, _, _ = _pmc_load(("r06_topk_stream_pmc.json", "r05_topk_stream_pmc.json", "r04_topk_stream_pmc.json", "r03_topk_stream_pmc.json"))
    try:
        return rec["kernels"]["cosine_topk_stream_kernel<8>"]["traffic_bytes_per_launch"]
    except (TypeError, KeyError):
        return None


def parity_of_timed_mode(args, mo, cfg_o, st):
    """Checker leg (outside every timed region): the three operand formats -- fp16 (what the headline times by default), bf16, and
    the exact-fp32 parity mode -- on a B = 32 slice of the benchmark's synthetic batch against the CPU oracle with the same weights,
    noise and inputs."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    import numpy as np
    B = 32
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0)
    noise = torch.rand(B, 16, generator=g)
    loss_o, pred_o, mask_o, _, _, grads_o = mo.loss_and_grads(st, imgs, cfg_o, 0.75, noise)
    out = {"sample": "config A, B = 32 slice of the synthetic batch, seed-0 reference init, against oracle/mae_oracle.py (fp32, CPU)"}
    for name, dtype in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
        eng = MAEEngine(config_for("base", patch_size=16, in_chans=5, img_size=64, embed_dim=768), compute_dtype=dtype, seed=0)
        eng.load_state_dict(st)
        loss, pred, mask = eng.forward_train(imgs.cuda(), 0.75, noise.cuda())
        eng.backward()
        torch.cuda.synchronize()
        rel = lambda a, b: float(np.linalg.norm(a.double().numpy() - b.double().numpy()) / (np.linalg.norm(b.double().numpy()) + 1e-30))
        gr = max(rel(eng.grad(k).cpu().reshape(grads_o[k].shape), grads_o[k]) for k in eng.store.order)
        out[name] = dict(loss_rel=abs(float(loss) - float(loss_o)) / float(loss_o), pred_rel_l2=rel(pred.cpu(), pred_o),
                         grad_rel_l2_max=gr, mask_equal=bool(torch.equal(mask.cpu(), mask_o)))
        del eng
        torch.cuda.empty_cache()
    return out


def cpu_baselines(args, search_inputs):
    """The CPU restatement (oracle/) timed on this box's host cores, rank 0 at N = 1 only, on bounded samples of the SAME
    workloads: pretraining = config A at B = 256, one warm-up + three timed optimiser steps;
    search = Q = 1 and Q = 64 over the full bank, k as benchmarked."""
    from oracle import mae_oracle as mo
    from oracle import similarity_oracle as so
    threads = torch.get_num_threads()
    cfg = mo.config_for("base", patch_size=16, in_chans=5, img_size=64, embed_dim=768)
    st = mo.init_state(cfg, seed=0)
    parity = parity_of_timed_mode(args, mo, cfg, st)           # (before the oracle's optimiser steps move `st`)
    tr = mo.Trainer(cfg, st, init_lr=1e-4, weight_decay=0.05, total_iters=1_000_000, final_lr_factor=1e7)
    B = args.batch
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0)
    tr.step(imgs, 0.75, torch.rand(B, 16, generator=g))                       # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for _ in range(3):
        tr.step(imgs, 0.75, torch.rand(B, 16, generator=g))
    dt = (time.perf_counter() - t0) / 3
    pre = dict(value=B / dt, unit="images/sec", cores=threads, kind="port",
               sample=f"config A at B={B}: 1 warm-up + 3 timed optimiser steps (fwd+bwd+AdamW), torch fp32 CPU restatement "
                      f"(oracle/mae_oracle.py), {dt:.2f} s per step")
    pre["parity"] = parity
    sea = None
    if search_inputs is not None:
        queries, w, bank = search_inputs
        x = bank.cpu().numpy()
        q = queries[:64].cpu().numpy()
        wh = w.cpu().numpy()
        so.cosine_topk_np(q[:1], x[:1000], 10, wh)                           # loads the library, spins up OpenMP
        legs = {}
        for Q in (1, 64):
            t0 = time.perf_counter()
            so.cosine_topk_np(q[:Q], x, args.topk, wh)
            legs[Q] = time.perf_counter() - t0
        sea = dict(value=64 / legs[64], unit="queries/sec", cores=so.num_threads(), kind="port",
                   sample=f"Q=64, k={args.topk} over the full {x.shape[0]}x{x.shape[1]} bank (oracle/topk_oracle.c, OpenMP): "
                          f"{legs[64]:.2f} s; Q=1: {legs[1]:.2f} s = {1 / legs[1]:.2f} queries/sec",
                   q1_queries_per_sec=1 / legs[1])
    return pre, sea


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run, as the driver does.
    Nothing in this process has touched the GPU (device_count() does not initialise it on this image)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("SKYEMB_BENCH_REHEARSAL", "0") != "1":
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible on this node", file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.skip_leg:
        return skip_leg(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # SKYEMB_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- walks the N > 1 code path of this file on a one-GPU box (the
    # numbers mean nothing: the ranks share the card and the collectives go through the host; the line says so)
    rehearsal = world > 1 and os.environ.get("SKYEMB_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    pre, eng = bench_pretrain(args, rank, world, dev)
    del eng
    torch.cuda.empty_cache()
    search = search_inputs = None
    if not args.skip_search:
        search, search_inputs = bench_search(args, rank, world, dev)
    mim19 = None
    if rank == 0 and world == 1 and not args.skip_mim19:
        mim19 = bench_mim19(args, dev)
    if rank == 0:
        executed = pre["flops_per_image_executed"]
        ach = pre["images_per_sec"] / world * executed / 1e12   # per-GPU TFLOP/s, executed FLOPs
        peak = PEAK_BF16_TFLOPS if args.dtype in ("bf16", "f16") else PEAK_F32_MFMA_TFLOPS
        gi = pre["gemm_in_step"]
        kernel = dict(kernel="gemm_pipe_kernel<BM,BN,A_KC,B_KC,NSTAGE,WM,WN,WK> (forward KC.KC, data gradient KC.RC; WK = 2 k-groups of waves "
                             "on the <= 256-tile launches) + gemm_pipe_group_kernel<128,128,2,4,2,4> (encoder) / <128,64,3,4,2,4> (decoder): the "
                             "four weight gradients of a block per launch (+ splitk_reduce_kernel)", **gi)
        if "gemm_probe" in pre:
            kernel["isolated_probe"] = pre["gemm_probe"]
        if pre.get("fused_adamw") and "staged" in pre and "monolithic_separate_adamw" in pre["staged"]:
            sep = pre["staged"]["monolithic_separate_adamw"]
            if "gemm_ms_per_step" in sep:
                kernel["with_separate_adamw_launch"] = dict(ms_per_step=sep["gemm_ms_per_step"], tflops=gi["flop_per_step"] / sep["gemm_ms_per_step"] / 1e9,
                                                            frac=gi["flop_per_step"] / sep["gemm_ms_per_step"] / 1e9 / peak,
                                                            note="the same GEMM family in the step that keeps AdamW as its own launch "
                                                                 "(no optimiser traffic in the weight-gradient epilogues)")
        line = {
            "metric": "MAE pretrain images/sec (5x64x64, ViT-B)", "value": pre["images_per_sec"], "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": pre["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "mim_32.ini-as-BASELINE configs[1]: MAE ViT-Base/16, 5x64x64, mask_ratio=0.75, "
                                   f"bs={pre['B']}/GPU, AdamW+cosine, norm_pix mse",
                       "global_batch": pre["B"] * world, "parallelism": f"dp{world}",
                       "operand_format": {"f16": "IEEE half MFMA operands (v_mfma_f32_16x16x32_f16; same dense peak as bf16), fp32 accumulation / "
                                                 "statistics / residual stream / master weights / optimiser state, static loss scale 2^16: the 16-bit "
                                                 "mode inside the reference tolerance (parity.f16); BASELINE names bf16: extra.operand_format_ab "
                                                 "times both, interleaved",
                                          "bf16": "bf16 MFMA operands as BASELINE configs[1] names (parity.bf16: 6e-3 on the pixels)",
                                          "f32": "exact fp32 MFMA chains"}[args.dtype],
                       "rccl_ranks": 0 if rehearsal else torch.distributed.get_world_size() if world > 1 else 1,
                       **({"rehearsal": "SKYEMB_BENCH_REHEARSAL=1: all ranks on cuda:0 over gloo; a walk through the N > 1 code path, "
                                        "NOT a measurement"} if rehearsal else {}),
                       "graph": not args.no_graph, "adamw_in_wgrad_epilogue": bool(pre.get("fused_adamw")),
                       "adamw_placement": (f"policy {pre.get('adamw_side')}: epilogue of each block's own grouped weight-gradient launch at this size "
                                           "(side jobs of the next block's launch where launches leave compute units idle: ViT-L, extra.mim_19)"
                                           if pre.get("fused_adamw") else "separate launch")},
            "roofline": {"bound": "mfma", "achieved": gi["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": gi["tflops"] / peak,
                         # the same family with AdamW as its own launch (no optimiser bytes in the weight-gradient epilogues): the figure
                         # comparable with rounds 1-2, whose `frac` had no optimiser traffic in it
                         "frac_gemm_only": kernel.get("with_separate_adamw_launch", {}).get("frac"),
                         "traffic": gemm_pmc_traffic(),
                         **traffic_ratio(),
                         "note": "(one process: the AdamW step of the transformer blocks' weights runs INSIDE the grouped weight-gradient "
                                 "launches, so the family's in-step time includes that HBM traffic -- 26 B per parameter; "
                                 "extra.staged.monolithic_separate_adamw is the step with the separate optimiser launch) "
                                 "dominant kernel family = the pipelined MFMA GEMM launches: algorithmic 2MNK FLOPs of every GEMM "
                                 "launch of one step / their IN-STEP time = HIP-event time of the timed steps minus the same "
                                 "steps replayed with the GEMM launches left out (kernel.step_without_gemm_ms); agrees with the "
                                 "sum of the gemm_pipe_* rows of the rocprofv3 kernel_stats in profiles/.  step = whole-step rate "
                                 "(executed fwd+bwd FLOPs x images / step time, all kernels + optimiser)",
                         "kernel": kernel,
                         "step": {"achieved": ach, "frac": ach / peak, "gpu_ms_per_step": pre["gpu_ms_per_step"],
                                  "flops_per_image_executed": executed,
                                  "flops_per_image_reference": pre["flops_per_image_reference"]}},
            "loss": pre["loss"],
            "phases": pre.get("phases"),
            "feeder": pre.get("feeder"),
        }
        line["extra"] = {}
        if mim19 is not None:
            line["extra"]["mim_19"] = mim19
        for key in ("long_run", "staged", "batch_sweep", "optimizer_placement", "operand_format_ab"):
            if key in pre:
                line["extra"][key] = pre[key]
        if world == 1 and args.dtype in ("bf16", "f16") and not args.skip_f32:
            line["extra"]["f32_mode"] = f32_mode_timing(dev, rank, pre["B"])
        if search is not None:
            ql, qs = search["q_large"], search["q_small"]
            line["search"] = {
                "metric": f"cosine top-k queries/sec over {args.bank_rows}x768", "value": ql["queries_per_sec"],
                "unit": "queries/sec", "k": args.topk, "Q": ql["Q"], "dtype": "f32", "sharding": f"bank rows / {world}",
                "q_large": ql, "q_small": qs, "q1": search["q1"],
                "roofline": {"bound": "hbm", "achieved": qs["kernel_hbm_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": qs["kernel_hbm_gbs"] / PEAK_HBM_GBS, "traffic": pmc_traffic(args),
                             "algorithmic_bytes": qs["kernel_bytes"],
                             # the whole search as the caller sees it: query preparation, sample floor, bank pass, merge
                             "frac_end_to_end": qs["kernel_bytes"] / qs["sec"] / 1e9 / PEAK_HBM_GBS,
                             "frac_end_to_end_q1": search["q1"]["kernel_bytes"] / search["q1"]["sec"] / 1e9 / PEAK_HBM_GBS,
                             "note": "Q=16 bank-streaming launch (HBM-bound regime): (bank shard + queries + partial "
                                     "lists) bytes / kernel time; the Q=10k path: roofline_q_large"},
                # BASELINE's own search config (10 000 queries: compute-bound): 2 Q N D algorithmic FLOPs over the END-TO-END time
                # of the search (fp16 image of the queries, matrix pass, candidate selection, exact re-score), against the dense
                # fp16 MFMA peak the first stage runs on
                "roofline_q_large": {"bound": "mfma_f16", "achieved": ql["effective_tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                     "frac": ql["effective_tflops"] / PEAK_BF16_TFLOPS,
                                     "note": "two-stage exact search (csrc/topk_prefilter.hip); the exact single-stage fp32 kernel on the "
                                             "same inputs: q_large.kernel_tflops of the 157 TFLOP/s fp32-MFMA peak",
                                     **search_pmc_record()},
            }
        if world == 1 and not args.skip_cpu:
            cpre, csea = cpu_baselines(args, search_inputs)
            line["parity"] = cpre.pop("parity")
            line["parity"]["timed_mode"] = args.dtype
            line["parity"]["note"] = ("north_star: loss and reconstructed pixels within 1e-3 relative of the fp32 reference.  f16 (IEEE half operands, "
                                      "the default timed mode): inside it (pred_rel_l2 ~7e-4, loss ~1e-5; tests/test_f16_gpu.py holds B = 256 to 1e-3); "
                                      "bf16 (BASELINE's named dtype, same kernels and rate): 6e-3 on the pixels -- the rounding of 8-bit "
                                      "significands, measured operand class by operand class in profiles/r06_operand_rounding.json; f32 (exact "
                                      "fp32 MFMA chains): ~1e-6")
            pt = line["parity"].get("f16")
            if pt is not None and "operand_format_ab" in line["extra"]:
                ab = line["extra"]["operand_format_ab"]
                line["extra"]["tol_mode"] = dict(dtype="f16", ms_per_step=ab["mean_ms"]["f16"], images_per_sec=ab["images_per_sec"]["f16"],
                                                 pred_rel_l2=pt["pred_rel_l2"], loss_rel=pt["loss_rel"], grad_rel_l2_max=pt["grad_rel_l2_max"],
                                                 is_headline=args.dtype == "f16",
                                                 note="the 16-bit mode that meets the reference tolerance (1e-3 on loss and pixels)")
            line["cpu_baseline"] = cpre
            if search is not None and csea is not None:
                line["search"]["cpu_baseline"] = csea
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _pmc_load(names):
    """(record, file, current) of the newest committed PMC summary among `names`; current = its `csrc_sha16` stamp equals the
    fingerprint of the kernel sources this process runs (tools/fingerprint.py) -- hardware-counter figures of older kernels are
    reported with pmc_current = false instead of sitting beside fresh timings as if they were theirs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from fingerprint import csrc_sha16
        now = csrc_sha16(ROOT)
    except Exception:
        now = None
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            return rec, "profiles/" + name, bool(now and rec.get("csrc_sha16") == now)
        except (OSError, ValueError):
            continue
    return None, None, False


def mim19_pmc_record():
    """Matrix-pipe utilisation of the mim_19 step's GEMM kernels by hardware counters (profiles/r04_mim19_pmc.json: cycle-weighted
    SQ_VALU_MFMA_BUSY_CYCLES over the launches' SIMD-cycles, a separate --pmc pass over two eager steps), or {}."""
    try:
        rec, src, cur = _pmc_load(("r06_mim19_pmc.json", "r05_mim19_pmc.json", "r04_mim19_pmc.json"))
        ks = {n: v for n, v in rec["kernels"].items() if n.startswith("gemm")}
        cyc = {n: v["gpu_cycles_per_launch"] * v["launches"] for n, v in ks.items()}
        tot = sum(cyc.values())
        return {"gemm_mfma_busy_pmc": sum(ks[n].get("mfma_util", 0.0) * cyc[n] for n in ks) / tot,
                "gemm_share_of_gpu_cycles_pmc": sum(v["share_of_gpu_cycles"] for v in ks.values()), "pmc_source": src, "pmc_current": cur}
    except (TypeError, KeyError, ValueError, ZeroDivisionError):
        return {}


def search_pmc_record():
    """Matrix-pipe utilisation of the many-query pass by hardware counters (profiles/r04_search_pmc.json: SQ_VALU_MFMA_BUSY_CYCLES over
    the launch's SIMD-cycles, a separate --pmc pass), or {}."""
    try:
        rec, src, cur = _pmc_load(("r06_search_pmc.json", "r05_search_pmc.json", "r04_search_pmc.json"))
        ks = rec["kernels"]
        k = next(v for n, v in ks.items() if n.startswith("prefilter_kernelILi2ELb0"))
        return {"mfma_busy_pmc": k["mfma_util"], "l2_hit_rate_pmc": k["l2_hit_rate"], "pmc_source": src, "pmc_current": cur,
                "pmc_note": "prefilter_kernel<2,false> (80 % of the search): matrix pipe busy / SIMD-cycles of the launch at the clock the "
                            "chip ran it at (1.75 GHz in the two long launches: 12.3 / 14.3 M cycles in 6.9 / 8.0 ms); `frac` prices the FLOPs against the 2.4 GHz peak"}
    except (TypeError, KeyError, ValueError, StopIteration):
        return {}


def gemm_pmc_record():
    """The GEMM family's record of the newest committed PMC summary (profiles/rNN_mfma_pmc.json: FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE, separate --pmc passes), or None."""
    rec, src, cur = _pmc_load(("r06_mfma_pmc.json", "r05_mfma_pmc.json", "r04_mfma_pmc.json", "r03_mfma_pmc.json"))
    try:
        rec = dict(rec["gemm_family_total"])
    except (TypeError, KeyError):
        return None
    rec["file"], rec["current"] = src, cur
    return rec


def traffic_ratio():
    """{traffic_ratio, traffic_algorithmic}: PMC bytes over the algorithmic bytes of the GEMM family (each operand and output once +
    the optimiser state the fused weight-gradient epilogues move), per launch like `traffic`."""
    rec = gemm_pmc_record()
    if not rec or "algorithmic_bytes_per_step_gemm_operands" not in rec:
        return {}
    alg = rec["algorithmic_bytes_per_step_gemm_operands"] + rec.get("optimizer_bytes_in_weight_gradient_epilogues", 0)
    return {"traffic_algorithmic": alg / rec["launches_per_step"], "traffic_ratio": rec["hbm_side_bytes_per_step"] / alg,
            "traffic_source": rec["file"], "traffic_pmc_current": rec["current"]}


def gemm_pmc_traffic():
    """HBM-side bytes per GEMM launch (average over the launches of one step), or None."""
    rec = gemm_pmc_record()
    return rec["hbm_side_bytes_per_launch"] if rec else None


if __name__ == "__main__":
    main()
