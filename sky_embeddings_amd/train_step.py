"""One optimiser step of MAE pretraining as a replayable HIP graph.

A step at BASELINE config A is ~500 short kernels (1.36 TFLOP in total): eager launches are
host-bound, so forward + backward are captured ONCE into a HIP graph (through torch's stream
capture; the kernels are this package's, launched via the C ABI on the capture stream) and
replayed per step.  The gradient all-reduce (RCCL, one process per GPU) and the single fused
AdamW launch follow the replay on the same stream.
"""
from __future__ import annotations

import torch

from .optim import CosineLR, FusedAdamW


class TrainStep:
    def __init__(self, engine, optimizer: FusedAdamW, scheduler: CosineLR, batch_size: int, mask_ratio: float = 0.75,
                 use_graph: bool = True, process_group=None, world_size: int = 1, warmup_iters: int = 2):
        self.engine, self.optimizer, self.scheduler = engine, optimizer, scheduler
        self.mask_ratio = mask_ratio
        self.world_size = world_size
        self.process_group = process_group
        cfg = engine.cfg
        dev = engine.device
        self.imgs = torch.zeros(batch_size, cfg.in_chans, cfg.img_size, cfg.img_size, device=dev)
        self.noise = torch.zeros(batch_size, cfg.num_patches, device=dev)
        self.graph = None
        self.loss = None
        if world_size > 1:
            optimizer.grad_scale = 1.0 / world_size  # DDP mean of per-rank gradients
        if use_graph:
            # warm up on a side stream (lazy hipFuncSetAttribute calls, workspace allocation), then capture
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(warmup_iters):
                    self._fwd_bwd()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._fwd_bwd()

    def _fwd_bwd(self):
        # utils/mim_vit.py:363 draws the masking noise inside forward; keep it inside the step
        self.noise.uniform_()
        self.loss, self.pred, self.mask = self.engine.forward_train(self.imgs, self.mask_ratio, self.noise)
        self.engine.backward()

    def load_batch(self, imgs):
        """Stage the next minibatch (device or pinned host tensor) into the static input buffer."""
        self.imgs.copy_(imgs, non_blocking=True)

    def __call__(self, imgs=None):
        if imgs is not None:
            self.load_batch(imgs)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._fwd_bwd()
        if self.world_size > 1:
            from .distributed import allreduce_flat_gradients
            allreduce_flat_gradients(self.engine.store.g, self.world_size, group=self.process_group)
        self.optimizer.step()
        self.scheduler.step()
        return self.loss
