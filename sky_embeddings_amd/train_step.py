"""One optimiser step of MAE pretraining as replayable HIP graphs.

A step at BASELINE config A is ~500 short kernels (1.34 TFLOP in total): eager launches are
host-bound, so the launch schedule is captured ONCE into HIP graphs (through torch's stream
capture; the kernels are this package's, launched via the C ABI on the capture stream) and
replayed per step.

* 1 GPU: one graph = noise draw + forward + backward; then the single fused AdamW launch.
* N GPUs (one process each): backward is cut into stages (decoder, encoder block groups, patch
  embedding).  After each stage's graph is enqueued, the slices of the flat gradient buffer it has
  finalised are all-reduced asynchronously (RCCL runs on its own stream, ordered after that stage),
  so communication overlaps the remaining backward stages; AdamW (grad_scale = 1/N) follows the
  last collective.  No other collective is used.
  Gradient communication dtype (``grad_comm`` / SKYEMB_GRAD_COMM): "bf16" (default with N > 1) -- the last
  kernel of each stage's graph casts the stage's slices into a flat bf16 buffer, the all-reduce sums THAT
  (225 MB per step at ViT-B instead of 449 MB: xGMI rings are per-link bound) and AdamW reads the bf16 sums;
  "f32" -- the fp32 buffer itself is all-reduced (bit-for-bit the mean of the ranks' fp32 gradients up to
  the reduction order).
"""
from __future__ import annotations

import os

import torch

from . import ops
from .distributed import all_gather_range, bucket_bounds, reduce_scatter_range
from .optim import CosineLR, FusedAdamW


def _subtract_ranges(ranges, holes):
    """[(s, e), ...] minus [(s, e), ...] (both ascending, disjoint) -> what remains, ascending."""
    out = []
    for s, e in ranges:
        cur = s
        for hs, he in holes:
            if he <= cur or hs >= e:
                continue
            if hs > cur:
                out.append((cur, hs))
            cur = max(cur, he)
        if cur < e:
            out.append((cur, e))
    return out


class TrainStep:
    def __init__(self, engine, optimizer: FusedAdamW, scheduler: CosineLR, batch_size: int, mask_ratio: float = 0.75,
                 use_graph: bool = True, process_group=None, world_size: int = 1, warmup_iters: int = 2,
                 staged: bool | None = None, n_encoder_groups: int | None = None, bucket_elems: int = 32 * 1024 * 1024,
                 wgrad_overlap: bool | None = None, optimizer_overlap: bool | None = None, grad_comm: str | None = None,
                 external_noise: bool = False, max_mask_ratio: float | None = None, fused_adamw: bool | None = None,
                 adamw_side: bool | None = None, shard_optimizer: bool | None = None, shard_emulate: tuple | None = None):
        self.engine, self.optimizer, self.scheduler = engine, optimizer, scheduler
        self.mask_ratio = mask_ratio
        self.world_size = world_size
        self.process_group = process_group
        # SKYEMB_DIST_FORCE=1: issue the per-stage collectives with ONE rank too (an initialised process group of size 1): the RCCL
        # calls, their stream ordering against the stage graphs and the optimiser stream, on a box with one GPU (tests/test_ddp_gpu.py)
        self.collectives = world_size > 1 or (os.environ.get("SKYEMB_DIST_FORCE", "0") == "1" and torch.distributed.is_available()
                                              and torch.distributed.is_initialized())
        self.bucket_elems = bucket_elems
        cfg = engine.cfg
        dev = engine.device
        self.imgs = torch.zeros(batch_size, cfg.in_chans, cfg.img_size, cfg.img_size, device=dev)
        self.noise = torch.zeros(batch_size, cfg.num_patches, device=dev)
        # SimMIM mode (SimMIMEngine): the per-pixel mask and the RA/Dec pairs are step inputs too
        self.simmim = bool(cfg.simmim)
        self.pixel_mask = torch.zeros_like(self.imgs) if self.simmim else None
        # SimMIM masks generated on the device inside the step (utils/dataloaders.py:197-219 moved out of the loader workers):
        # with max_mask_ratio set, callers hand over cutouts (+ RA/Dec) only
        self.max_mask_ratio = max_mask_ratio if self.simmim else None
        if self.max_mask_ratio is not None:
            self.mask_noise = torch.zeros(batch_size, cfg.in_chans, cfg.num_patches, device=dev)
            self.ratio_u = torch.zeros(batch_size, device=dev)
        self.ra_dec = torch.zeros(batch_size, 2, device=dev) if cfg.ra_dec else None
        self.loss = None
        self.external_noise = external_noise          # parity tests fill step.noise themselves (same noise on 1 and N ranks)
        explicit_comm = grad_comm is not None
        if grad_comm is None:
            grad_comm = os.environ.get("SKYEMB_GRAD_COMM", "bf16")
        assert grad_comm in ("bf16", "f32"), grad_comm
        # (one process: no communication, so no mirror -- unless the caller asks for the N > 1 schedule explicitly, e.g.
        # bench.py pricing the stage graphs + casts of the data-parallel step on one GPU)
        self.grad_comm = grad_comm if (world_size > 1 or (staged and explicit_comm)) else "f32"
        self.g16 = None
        if self.grad_comm == "bf16":
            # (the mirror holds the compute dtype's 16-bit format: the weight-gradient launches write it in their epilogue.  In the
            # fp16 mode the sums carry the loss scale: world_size x 2^16 x a gradient element stays far inside fp16's range)
            self.g16 = torch.zeros(engine.store.n, device=dev, dtype=engine.dtype if engine.dtype in ops.LP_DTYPES else torch.bfloat16)
        if staged is None:
            env = os.environ.get("SKYEMB_STAGED")
            staged = (world_size > 1) if env is None else env == "1"
        self.staged = staged
        if wgrad_overlap is None:
            # measured slower on one MI355X (8.96 vs 8.47 ms/step: the cross-branch graph edges cost more than the
            # bubbles they fill), so off unless asked for
            wgrad_overlap = os.environ.get("SKYEMB_WGRAD_OVERLAP", "0") == "1"
        engine.enable_wgrad_overlap(wgrad_overlap)   # weight-gradient GEMMs on a side stream (a parallel graph branch)
        # DDP mean of per-rank gradients, and the backward pass's static loss scale divided out again (fp16 mode)
        if hasattr(engine, "plan_loss_scale"):     # fp16 mode: the scale of THIS batch size, before anything bakes it into a launch
            engine.plan_loss_scale(engine.expected_masked_elements(batch_size, mask_ratio))
        optimizer.base_grad_scale = 1.0 / world_size
        # stage list: [(callable, [(start, end) slices of the flat gradient buffer final after it])]
        if n_encoder_groups is None:
            # finer stages with N GPUs: the all-reduce of the LAST encoder group has only the short embedding stage to hide
            # behind, so that group is kept small (2 of 12 blocks at ViT-B: 28 MB of bf16 gradients)
            n_encoder_groups = 6 if world_size > 1 else 3
        if self.staged:
            stages = engine.backward_stages(min(n_encoder_groups, cfg.depth))
            first_fn, first_ranges = stages[0]
            self.stages = [((lambda: (self._forward(), first_fn())), first_ranges)] + stages[1:]
        else:
            self.stages = [((lambda: (self._forward(), engine.backward())), [(0, engine.store.n)])]
        if self.g16 is not None:
            # bf16 gradient communication: the blocks' grouped weight-gradient launches write their bf16 gradients straight
            # into the mirror (98 % of the bytes: no fp32 store + cast pass for them); each stage ends with the cast of whatever
            # else it has finalised (embeddings, biases, LayerNorms, the three single weight gradients)
            direct = os.environ.get("SKYEMB_G16_DIRECT", "1") == "1" and hasattr(engine, "enable_grad_mirror")
            if direct:
                engine.enable_grad_mirror(self.g16)

            def with_cast(fn, ranges):
                todo = []

                def run():
                    engine._g16_active = direct
                    try:
                        fn()
                    finally:
                        engine._g16_active = False
                    if not todo:                              # (the first call has just built the workspace and its groups)
                        holes = []
                        if direct:
                            holes = engine.grad_mirror_ranges(engine._ws[engine._last_key()])   # the workspace this step ran on
                        todo.append(_subtract_ranges(ranges, holes))
                    for (s, e) in todo[0]:
                        ops.cast(engine.store.g[s:e], self.g16[s:e], e - s)
                return run
            self.stages = [(with_cast(fn, ranges), ranges) for fn, ranges in self.stages]
        # optimiser overlap: AdamW of a stage's slices runs on a side stream as soon as that stage (and its all-reduce)
        # is done, concurrently with the remaining backward stages -- an HBM-bound kernel next to L2/LDS-bound GEMMs
        if optimizer_overlap is None:
            # measured on one MI355X: 6.76 ms/step with the overlap vs 6.61 without (the HBM-bound update slows the
            # concurrent GEMMs by more than it hides), so it is opt-in; with N GPUs it can fill all-reduce waits
            optimizer_overlap = self.staged and os.environ.get("SKYEMB_OPT_OVERLAP", "0") == "1"
        self.optimizer_overlap = bool(optimizer_overlap and self.staged)
        if self.optimizer_overlap:
            covered = sorted(r for _, rs in self.stages for r in rs)
            assert covered[0][0] == 0 and covered[-1][1] == engine.store.n and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), \
                "backward stages must partition the flat parameter buffer"
            self.opt_stream = torch.cuda.Stream(device=dev)
        # N > 1: the optimiser SHARDED over the ranks (round 6).  The replicated schedule has every rank step all 112 M parameters
        # (3.4 GB of optimiser traffic per rank and step: 0.66 ms of a 4.8 ms step) after an all-reduce.  Here every stage range of
        # the transformer blocks' weights is cut into `world` equal chunks: reduce-scatter (this rank receives the summed gradients of
        # ITS chunk only) -> AdamW on that chunk of p / m / v (1 / world of the traffic) -> all-gather of the 16-bit shadow the GEMMs
        # read.  xGMI bytes are those of the all-reduce (which IS a reduce-scatter + an all-gather).  The fp32 master weights and
        # moments of a chunk are current on its owner only: gather_full_state() collects them before a checkpoint.  The last
        # stage's ranges (embeddings, biases, LayerNorms: 2 % of the elements, read by the forward pass as fp32) stay replicated.
        # Results equal the replicated schedule (same sums in a possibly different order).  shard_emulate = (rank, world): the
        # sharded schedule's COMPUTE on one process, no collectives -- bench.py's pricing leg; parameters are then wrong by design.
        if shard_optimizer is None:
            env = os.environ.get("SKYEMB_SHARD_OPT", "auto")
            shard_optimizer = (world_size > 1) if env == "auto" else env == "1"
        self.shard_rank, self.shard_world = (shard_emulate if shard_emulate is not None else
                                             ((torch.distributed.get_rank(process_group), world_size) if self.collectives and world_size > 1 else (0, 1)))
        self.shard_emulated = shard_emulate is not None
        # (one rank with the collectives forced -- SKYEMB_DIST_FORCE=1, the one-GPU RCCL test -- shards over a world of one when asked
        # to explicitly: reduce_scatter_tensor / all_gather_into_tensor of a single rank are copies, issued for real)
        forced_one = self.collectives and self.shard_world == 1 and shard_optimizer is True and os.environ.get("SKYEMB_SHARD_OPT", "auto") != "0"
        self.shard_optimizer = bool(shard_optimizer and self.staged and (self.shard_world > 1 or forced_one) and not self.optimizer_overlap)
        self._own = {}
        if self.shard_optimizer:
            from .distributed import shard_chunk
            gdt = engine.store.g.dtype if self.g16 is None else self.g16.dtype
            for k, (_, ranges) in enumerate(self.stages[:-1]):
                for (s_, e_) in ranges:
                    self._own[(s_, e_)] = torch.zeros(max(shard_chunk(e_ - s_, self.shard_world), 8), device=dev, dtype=gdt)
        # One process per replica: the AdamW step of every transformer block's weights runs in the epilogue of that block's grouped
        # weight-gradient launch (no gradient round trip through HBM, no separate pass over 99 % of the parameters); the ordinary
        # kernel updates the rest after the graph.  Not with N > 1 (the all-reduce sits between backward and the update).
        if fused_adamw is None:
            fused_adamw = os.environ.get("SKYEMB_FUSED_ADAMW", "1") == "1"
        self.fused_adamw = bool(fused_adamw and world_size == 1 and not self.staged and not self.optimizer_overlap
                                and engine.dtype in ops.LP_DTYPES and hasattr(engine, "enable_fused_adamw"))
        self._rest_ranges = None
        snap = None
        if self.fused_adamw:
            optimizer.use_device_scalars(dev)
            # (adamw_side: the step of a block's weights as a side job of the NEXT block's weight-gradient launch instead of its own
            # launch's epilogue -- engine.enable_fused_adamw; None = the default, SKYEMB_ADAMW_SIDE)
            engine.enable_fused_adamw(optimizer, side=adamw_side)
            self.adamw_side = engine._adamw_side          # placement policy: 'auto' | '0' | '1' | 'dec' | 'enc'
            # (the warm-up launches below would already step the fused tensors: restore them afterwards)
            st = engine.store
            snap = [t.clone() for t in (st.p, st.m, st.v, st.p_lp)]

            # what the fused launches were built with: changing these afterwards (load_state_dict, a new grad_scale) would leave
            # the block weights stepping with the old constants while apply_range() uses the new ones
            self._fused_consts = self._optimizer_consts()

            def owning(fn):
                def run():
                    engine._fused_active = True          # these launches carry the optimiser step
                    try:
                        fn()
                    finally:
                        engine._fused_active = False
                    # ... and the ordinary kernel updates the rest (embeddings, biases, LayerNorms, the single weight gradients)
                    # right behind them, inside the same graph: as separate launches after the replay they started 8.6 us late
                    # (the host's launch latency on the critical path of every step)
                    for (s_, e_) in self._rest():
                        optimizer.apply_range(s_, e_)
                return run
            self.stages = [(owning(fn), ranges) for fn, ranges in self.stages]
        self.graphs = None
        if use_graph:
            # warm up on a side stream (lazy hipFuncSetAttribute calls, workspace allocation), then capture
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(warmup_iters):
                    for fn, _ in self.stages:
                        fn()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graphs = []
            pool = None
            for fn, _ in self.stages:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    fn()
                pool = g.pool()
                self.graphs.append(g)
            engine._graph_captures = getattr(engine, "_graph_captures", 0) + 1     # (engine.enable_wgrad_overlap refuses to re-plan under them)
        if snap is not None:
            if not use_graph:                                  # (graph mode ran the warm-up above: the workspace exists)
                self.stages[0][0]()
            torch.cuda.synchronize(dev)
            for dst, src in zip((engine.store.p, engine.store.m, engine.store.v, engine.store.p_lp), snap):
                dst.copy_(src)

    def _sharded_update(self):
        """AdamW on this rank's chunk of every sharded range (gradients: the reduce-scatter's output) and on the replicated rest,
        then the all-gather of the 16-bit shadow; the next forward waits for it on the stream."""
        from .distributed import shard_chunk
        opt, st = self.optimizer, self.engine.store
        r, W = self.shard_rank, self.shard_world
        opt.begin_step()
        gathers = []
        for k, (_, ranges) in enumerate(self.stages):
            for (s, e) in ranges:
                if k + 1 == len(self.stages):
                    opt.apply_range(s, e)                                  # replicated: embeddings, biases, LayerNorms
                    continue
                c = shard_chunk(e - s, W)
                if c > 0:
                    own = self._own[(s, e)]
                    if self.shard_emulated:                                # (pricing leg: no collective filled `own`)
                        opt.apply_range(s + r * c, s + (r + 1) * c)
                    else:
                        opt.apply_range(s + r * c, s + (r + 1) * c, grad=own[:c])
                opt.apply_range(s + W * c, e)                              # the range's tail (< 8 W elements): replicated
                if self.collectives and c > 0:
                    gathers.append(all_gather_range(st.p_lp, s, e, r, W, self.process_group))
        for w in gathers:
            if w is not None:
                w.wait()

    def gather_full_state(self):
        """With the optimiser sharded, a chunk's fp32 master weights and moments are current on its owner only: collect them on every
        rank (all-gather of p, m, v over the sharded ranges) -- call on ALL ranks before state_dict() / a checkpoint.  No-op otherwise."""
        if not (self.shard_optimizer and self.collectives) or self.shard_emulated:
            return
        st = self.engine.store
        works = []
        for k, (_, ranges) in enumerate(self.stages[:-1]):
            for (s, e) in ranges:
                for buf in (st.p, st.m, st.v):
                    works.append(all_gather_range(buf, s, e, self.shard_rank, self.shard_world, self.process_group))
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize(self.engine.device)

    def _optimizer_consts(self):
        o = self.optimizer
        return (tuple(o.defaults["betas"]), o.defaults["eps"], o.param_groups[1]["weight_decay"], o.grad_scale)

    def _rest(self):
        """Slices of the flat buffers the fused launches of THIS step's workspace do not update (the ordinary kernel takes them)."""
        if self._rest_ranges is None:
            eng = self.engine
            w = eng._ws[eng._last_key()]                       # the workspace forward_train has just run on
            fused = eng.fused_adamw_ranges(w)
            bounds = [0] + [b for r in fused for b in r] + [eng.store.n]
            self._rest_ranges = [(bounds[i], bounds[i + 1]) for i in range(0, len(bounds), 2) if bounds[i] < bounds[i + 1]]
            self._rest_key = eng._last_key()
        assert self._rest_key == self.engine._last_key(), "TrainStep: the batch shape changed under a fused optimiser step"
        return self._rest_ranges

    def _forward(self):
        if self.simmim:
            if self.max_mask_ratio is not None:
                if not self.external_noise:
                    self.mask_noise.uniform_()
                    self.ratio_u.uniform_()
                cfg = self.engine.cfg
                ops.simmim_mask_from_noise(self.mask_noise, self.ratio_u, float(self.max_mask_ratio), cfg.grid, cfg.patch_size,
                                           self.pixel_mask)
            self.loss, self.pred, self.mask = self.engine.forward_train(self.imgs, mask=self.pixel_mask, ra_dec=self.ra_dec)
            return
        # utils/mim_vit.py:363 draws the masking noise inside forward; keep it inside the step
        if not self.external_noise:
            self.noise.uniform_()
        self.loss, self.pred, self.mask = self.engine.forward_train(self.imgs, self.mask_ratio, self.noise, ra_dec=self.ra_dec)

    def load_batch(self, imgs, mask=None, ra_dec=None):
        """Stage the next minibatch (device or pinned host tensors) into the static input buffers."""
        self.imgs.copy_(imgs, non_blocking=True)
        if self.simmim:
            if self.max_mask_ratio is None:
                assert mask is not None, "SimMIM steps need the per-pixel mask (or TrainStep(max_mask_ratio=...))"
                self.pixel_mask.copy_(mask, non_blocking=True)
        if self.ra_dec is not None:
            self.ra_dec.copy_(ra_dec, non_blocking=True)

    def __call__(self, imgs=None, mask=None, ra_dec=None):
        if imgs is not None:
            self.load_batch(imgs, mask, ra_dec)
        if self.fused_adamw:
            # step t's scalars to the device, forward + backward (+ the fused updates), the ordinary kernel on the rest
            if self._optimizer_consts() != self._fused_consts:
                raise RuntimeError("TrainStep: betas / eps / weight_decay / grad_scale of the optimiser changed after the fused "
                                   "weight-gradient launches were built (load the checkpoint BEFORE constructing TrainStep, or rebuild it)")
            self.optimizer.begin_step()                    # (writes step t's scalars to the device buffer)
            if self.graphs is not None:
                self.graphs[0].replay()                    # forward + backward + every AdamW launch
            else:
                self.stages[0][0]()
            self.scheduler.step()
            return self.loss
        works = []
        g = self.engine.store.g if self.g16 is None else self.g16
        self.optimizer.grad_buffer = self.g16      # (eager steps outside TrainStep keep reading the fp32 buffer)
        overlap = self.optimizer_overlap
        main = torch.cuda.current_stream(self.engine.device)
        if overlap:
            self.optimizer.begin_step()
            self.opt_stream.wait_stream(main)          # the previous step's readers of the parameters are ordered before
        for k, (fn, ranges) in enumerate(self.stages):
            if self.graphs is not None:
                self.graphs[k].replay()
            else:
                fn()
            stage_works = []
            sharded = self.shard_optimizer and k + 1 < len(self.stages)
            if self.collectives and sharded:
                for (s, e) in ranges:
                    stage_works += reduce_scatter_range(g, s, e, self.shard_rank, self.shard_world, self._own[(s, e)], self.process_group)
            elif self.collectives:
                for (s, e) in ranges:
                    for (bs, be) in bucket_bounds(e - s, self.bucket_elems):
                        stage_works.append(torch.distributed.all_reduce(g[s + bs:s + be], group=self.process_group,
                                                                        async_op=True))
            if overlap:
                done = torch.cuda.Event()
                done.record(main)
                with torch.cuda.stream(self.opt_stream):
                    self.opt_stream.wait_event(done)
                    for w in stage_works:
                        w.wait()                       # the side stream waits for this stage's collectives
                    for (s, e) in ranges:
                        self.optimizer.apply_range(s, e)
            else:
                works += stage_works
        if overlap:
            main.wait_stream(self.opt_stream)          # next forward reads the updated parameters
        else:
            for w in works:
                w.wait()   # makes the compute stream wait for the collectives (no host block with NCCL/RCCL)
            if self.shard_optimizer:
                self._sharded_update()
            else:
                self.optimizer.step()
        self.optimizer.grad_buffer = None
        self.scheduler.step()
        return self.loss
