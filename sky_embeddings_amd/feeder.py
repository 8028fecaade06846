"""Batched HDF5 -> HBM input feeder (reference: utils/dataloaders.py:134-153, 221-328).

The reference's ``H5Dataset.__getitem__`` opens the file and reads ONE cutout per python call inside DataLoader worker
processes, then collates and pins.  At tens of thousands of images per second per GPU that per-item path is the
bottleneck, so the MI355X-first feeder works a minibatch at a time:

    memory-mapped contiguous HDF5 dataset --(native threads, skyemb_gather_rows_host)--> pinned ring slot
        --(ONE async H2D copy on a copy stream)--> device staging --(skyemb_clip_crop)--> device batch [B,C,S,S]

A background thread keeps ``depth`` batches in flight; the consumer stream only waits on an event.  Sample semantics are
the dataset's (clip at pixel_min, NaN kept, centre crop, RA/Dec pairs; zeros mask in MAE mode -- utils/dataloaders.py:323);
shuffling is a fresh permutation per epoch, sharded ``order[rank::world_size]`` for one-process-per-GPU runs.
The per-item ``build_h5_dataloader`` mirror stays available for transforms / SimMIM mask generation.
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from . import hdf5_lite
from ._lib import check, lib


def batches_per_epoch(n, batch_size, rank=0, world=1, drop_last=True):
    """Minibatches a rank draws per epoch from n cutouts: the same number on EVERY rank -- with several ranks the shards are cut
    to the common length n // world before batching (ranks take every world-th index, so low ranks would otherwise own one more)."""
    n_local = n // world if world > 1 else n
    return n_local // batch_size if (drop_last or world > 1) else (n_local + batch_size - 1) // batch_size


class CutoutFeeder:
    def __init__(self, path, batch_size, img_size=64, device="cuda", indices=None, shuffle=True, seed=0, pixel_min=-3.0,
                 pixel_max=None, depth=3, threads=4, drop_last=True, rank=0, world_size=1, epochs=1):
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the feeder targets device memory (there is no CPU path)"
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.f = hdf5_lite.File(path, "r")
        ds = self.f["cutouts"]
        assert ds.dtype == np.float32 and len(ds.shape) == 4, "cutouts must be float32 [N,C,H,W]"
        self.src = ds._array()                                   # np.memmap (contiguous dataset) -- never copied whole
        self.n_rows, self.C, self.Hs, self.Ws = ds.shape
        assert self.Hs >= img_size and self.Ws >= img_size
        self.row_bytes = self.C * self.Hs * self.Ws * 4
        self.ra = np.asarray(self.f["ra"], dtype=np.float32) if "ra" in self.f else np.zeros(self.n_rows, np.float32)
        self.dec = np.asarray(self.f["dec"], dtype=np.float32) if "dec" in self.f else np.zeros(self.n_rows, np.float32)
        self.indices = np.arange(self.n_rows, dtype=np.int64) if indices is None else np.asarray(indices, dtype=np.int64)
        self.B, self.S = int(batch_size), int(img_size)
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.rank, self.world = rank, world_size
        self.pixel_min, self.pixel_max = pixel_min, pixel_max
        self.depth, self.threads, self.epochs = max(2, depth), threads, epochs
        self.batch_size = self.B                                  # DataLoader-compatible attribute (pretrain_mim.py:93)
        self.dataset = self
        # every rank must run the SAME number of steps per epoch (each step issues collectives): with several ranks the
        # shards are cut to the common length len // world before batching (DistributedIndexSampler does the same)
        self._nb = batches_per_epoch(len(self.indices), self.B, self.rank, self.world, drop_last)
        # ring: pinned host slots, device staging + device output per slot
        self._pinned = [torch.empty(self.B, self.C, self.Hs, self.Ws, dtype=torch.float32).pin_memory() for _ in range(self.depth)]
        self._radec_pin = [torch.empty(self.B, 2, dtype=torch.float32).pin_memory() for _ in range(self.depth)]
        self._stage = [torch.empty(self.B, self.C, self.Hs, self.Ws, device=self.device) for _ in range(self.depth)]
        self._out = [torch.empty(self.B, self.C, self.S, self.S, device=self.device) for _ in range(self.depth)]
        self._radec = [torch.empty(self.B, 2, device=self.device) for _ in range(self.depth)]
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._zeros_mask = None

    def __len__(self):
        return self._nb

    def num_cutouts(self):
        return len(self.indices)

    def _epoch_order(self, epoch):
        idx = self.indices
        if self.shuffle:
            idx = idx[np.random.default_rng(self.seed + epoch).permutation(len(idx))]
        return np.ascontiguousarray(idx[self.rank::self.world])

    def _produce(self, q, free, stop):
        try:
            torch.cuda.set_device(self.device)
            for epoch in range(self.epochs):
                order = self._epoch_order(epoch)
                for b in range(self._nb):
                    if stop.is_set():
                        return
                    idx = np.ascontiguousarray(order[b * self.B:(b + 1) * self.B])
                    n = len(idx)
                    slot, ev = free.get()         # a ring slot the consumer has released ...
                    if slot is None:
                        return
                    if ev is not None:
                        ev.synchronize()          # ... and whose last use on the consumer's stream has finished
                    check(lib().skyemb_gather_rows_host(self.src.ctypes.data, self.row_bytes, idx.ctypes.data, n, self.n_rows,
                                                        self._pinned[slot].data_ptr(), self.threads), "skyemb_gather_rows_host")
                    rd = self._radec_pin[slot].numpy()
                    rd[:n, 0], rd[:n, 1] = self.ra[idx], self.dec[idx]
                    with torch.cuda.stream(self._copy_stream):
                        self._stage[slot][:n].copy_(self._pinned[slot][:n], non_blocking=True)
                        self._radec[slot][:n].copy_(self._radec_pin[slot][:n], non_blocking=True)
                        check(lib().skyemb_clip_crop(self._stage[slot].data_ptr(), self._out[slot].data_ptr(), n * self.C, self.Hs,
                                                     self.Ws, self.S, float(self.pixel_min if self.pixel_min is not None else 0.0),
                                                     float(self.pixel_max if self.pixel_max is not None else 0.0),
                                                     int(self.pixel_min is not None), int(self.pixel_max is not None),
                                                     self._copy_stream.cuda_stream), "skyemb_clip_crop")
                        ready = torch.cuda.Event()
                        ready.record(self._copy_stream)
                    q.put((slot, n, ready))
            q.put(None)
        except BaseException as e:   # surface producer failures in the consumer
            q.put(e)

    def __iter__(self):
        q, free = queue.Queue(), queue.Queue()
        for slot in range(self.depth):
            free.put((slot, None))
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, free, stop), daemon=True)
        th.start()
        prev = None
        try:
            while True:
                if prev is not None:               # the caller is done launching work on the previous batch
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    free.put((prev, ev))
                    prev = None
                while True:                        # a dead producer must not hang the training loop
                    try:
                        item = q.get(timeout=5.0)
                        break
                    except queue.Empty:
                        if not th.is_alive():
                            raise RuntimeError("CutoutFeeder: producer thread exited without a result")
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, n, ready = item
                torch.cuda.current_stream(self.device).wait_event(ready)
                if self._zeros_mask is None:
                    self._zeros_mask = torch.zeros(self.B, self.C, self.S, self.S, device=self.device)
                prev = slot
                yield self._out[slot][:n], self._zeros_mask[:n], self._radec[slot][:n]
        finally:
            stop.set()
            free.put((None, None))                 # wake a producer blocked on an empty free list
            th.join(timeout=10)
