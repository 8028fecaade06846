"""utils/dataloaders.py mirror: the HDF5 cutout path (H5Dataset, MaskGenerator, build_h5_dataloader, get_augmentations) and
the survey-tile path the reference's shipped MIM configs train from (``train_data_paths``: find_HSC_bands, load_fits_bands,
random / overlapping cutouts, FitsDataset, build_fits_dataloader).  The augmentation pipeline (utils/dataloaders.py:14-106)
is ``sky_embeddings_amd.augment.Augmenter``: torchvision's parameter draws on the host, the arithmetic in one HIP launch.
The tile sampler keeps the whole multi-band tile in HBM (the FITS bytes go to the device as they are) and cuts all of a
tile's windows with one launch; FITS access is ``sky_embeddings_amd.fits_lite`` (astropy is not a dependency).

Differences from the reference, none of which change a returned value:
  * the file is parsed once and datasets are memory-mapped (hdf5_lite) instead of re-opening
    the HDF5 file per item (utils/dataloaders.py:289);
  * ``open_h5`` is the single HDF5 access point (h5py is not required).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import hdf5_lite


def open_h5(path):
    return hdf5_lite.File(path, "r")


def extract_center(array, n):
    """utils/dataloaders.py: central n x n crop of a [C,H,W] (or [..,H,W]) array."""
    h, w = array.shape[-2:]
    top, left = (h - n) // 2, (w - n) // 2
    return array[..., top:top + n, left:left + n]


class MaskGenerator:
    """utils/dataloaders.py:155-219 -- SimMIM per-channel random patch masks (ratio = U(0,1)*max)."""

    def __init__(self, input_size=192, patch_size=4, max_mask_ratio=0.9, num_mask_chans=1):
        self.input_size, self.patch_size = input_size, patch_size
        self.max_mask_ratio, self.num_mask_chans = max_mask_ratio, num_mask_chans
        self.n_patches = self.input_size // self.patch_size
        self.token_count = self.n_patches ** 2

    def __call__(self):
        mask_ratio = torch.rand(1).item() * self.max_mask_ratio
        mask_count = int(torch.ceil(torch.tensor(self.token_count * mask_ratio)).item())
        masks = torch.zeros((self.num_mask_chans, self.token_count), dtype=torch.int)
        for i in range(self.num_mask_chans):
            masks[i, torch.randperm(self.token_count)[:mask_count]] = 1
        masks = masks.view(self.num_mask_chans, self.n_patches, self.n_patches)
        masks = masks.repeat_interleave(self.patch_size, dim=1).repeat_interleave(self.patch_size, dim=2)
        return masks.squeeze(0) if self.num_mask_chans == 1 else masks


class H5Dataset(torch.utils.data.Dataset):
    """utils/dataloaders.py:221-328: (cutout f32 [C,H,W] clipped at pixel_min, mask, ra_dec f32[2][, labels])."""

    def __init__(self, data_file, img_size, patch_size, num_channels, max_mask_ratio, num_patches=None, label_keys=None,
                 transform=None, pixel_min=-3., pixel_max=None, indices=None):
        self.data_file, self.transform, self.img_size = data_file, transform, img_size
        self.num_patches, self.label_keys = num_patches, label_keys
        self.pixel_min, self.pixel_max, self.indices = pixel_min, pixel_max, indices
        self.max_mask_ratio = max_mask_ratio
        self.mask_generator = (MaskGenerator(input_size=img_size, patch_size=patch_size, max_mask_ratio=max_mask_ratio,
                                             num_mask_chans=num_channels) if max_mask_ratio is not None else None)
        self._f = None

    def _file(self):
        if self._f is None:  # opened lazily so that DataLoader workers each map the file themselves
            self._f = open_h5(self.data_file)
        return self._f

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_f"] = None
        return d

    def __len__(self):
        if self.indices is not None:
            return len(self.indices)
        return len(self._file()['cutouts'])

    def __getitem__(self, idx):
        if self.indices is not None:
            idx = self.indices[idx]
        f = self._file()
        cutout = f['cutouts'][idx]
        if self.pixel_min is not None:
            cutout[cutout < self.pixel_min] = self.pixel_min      # NaN compares False: preserved
        if self.pixel_max is not None:
            cutout[cutout > self.pixel_max] = self.pixel_max
        if (np.array(cutout.shape[1:]) > self.img_size).any():
            cutout = np.ascontiguousarray(extract_center(cutout, self.img_size))
        ra_dec = torch.from_numpy(np.asarray([f['ra'][idx], f['dec'][idx]]).astype(np.float32))
        labels = None
        if self.label_keys is not None:
            labels = [f[k][idx] for k in self.label_keys]
            if 'class' in self.label_keys:
                labels = torch.from_numpy(np.asarray(labels).astype(np.int64)).long()
            else:
                labels = torch.from_numpy(np.asarray(labels).astype(np.float32))
        cutout = torch.from_numpy(cutout).to(torch.float32)
        if self.transform is not None:
            cutout = self.transform(cutout)
        mask = self.mask_generator() if self.mask_generator is not None else torch.zeros_like(cutout)
        if self.label_keys is None:
            return cutout, mask, ra_dec
        return cutout, mask, ra_dec, labels


def get_augmentations(img_size=64, flip=True, crop=True, brightness=0.8, noise=0.01, nan_channels=2):
    """utils/dataloaders.py:90-106: the target-augmentation pipeline; the returned object maps one [C, H, W] tensor to an
    augmented copy (like the reference's v2.Compose) and a whole batch to ``1 + A`` copies per sample with ``.batch``."""
    from ..augment import Augmenter
    return Augmenter(img_size=img_size, flip=flip, crop=crop, brightness=brightness, noise=noise, nan_channels=nan_channels)


def build_h5_dataloader(filename, batch_size, num_workers, patch_size=8, num_channels=5, max_mask_ratio=None,
                        label_keys=None, img_size=64, num_patches=None, augment=False, brightness=0.8, noise=0.01,
                        nan_channels=2, shuffle=True, indices=None, transforms=None, sampler=None):
    """utils/dataloaders.py:134-153 (+ optional ``sampler`` for one-process-per-GPU sharding)."""
    if (transforms is None) and augment:
        transforms = get_augmentations(img_size=img_size, brightness=brightness, noise=noise, nan_channels=nan_channels)
    dataset = H5Dataset(filename, img_size=img_size, patch_size=patch_size, num_channels=num_channels,
                        max_mask_ratio=max_mask_ratio, num_patches=num_patches, label_keys=label_keys,
                        transform=transforms, indices=indices)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle if sampler is None else False,
                                       sampler=sampler, num_workers=num_workers, pin_memory=torch.cuda.is_available())


# ---------------------------------------------------------------------------------------------------------------
# survey tiles (FITS): utils/dataloaders.py:330-654
# ---------------------------------------------------------------------------------------------------------------
def find_HSC_bands(fits_paths, bands, min_bands=2, verbose=1, use_calexp=True):
    """utils/dataloaders.py:330-379: [[file of band 0, file of band 1, ...], ...] per sky patch with at least ``min_bands`` of
    the bands present ('None' where a band is missing).  File names: ``[calexp-]<...>-<band>-<tract>-<patch>.fits``."""
    import glob
    import os
    per_patch = {}
    for root in fits_paths:
        for path in sorted(glob.glob(os.path.join(root, "*.fits"))):
            name = os.path.basename(path)
            if name.startswith("calexp-") != bool(use_calexp):
                continue
            parts = name.split("-")
            if len(parts) < 3 or parts[-3] not in bands:
                continue
            per_patch.setdefault("-".join(parts[-2:]), {})[parts[-3]] = path
    out = [[found.get(b, "None") for b in bands] for found in per_patch.values()
           if sum(b in found for b in bands) >= min_bands]
    if verbose:
        print(f"Found {len(out)} patches with at least {min_bands} of the {bands} bands.")
    return out


def load_fits_bands(patch_filenames, return_wc=False):
    """utils/dataloaders.py:381-447 on the host: -> (float array [C, H, W] with NaN planes for missing / unreadable bands,
    pix_to_radec or None).  (The sampler below does not go through this function: it sends the raw bytes to the GPU.)"""
    from .. import fits_lite
    planes, shape, pix_to_radec = [], None, None
    for fn in patch_filenames:
        img = None
        if fn != "None":
            try:
                hdu = fits_lite.read_image_hdu(fn, 1)
                img = hdu.array()
                shape = shape or img.shape
                if return_wc and pix_to_radec is None:
                    wcs = fits_lite.TanSipWCS(hdu.header)
                    pix_to_radec = lambda x, y, _w=wcs: _w.all_pix2world(x, y, 0)     # noqa: E731 (reference: dataloaders.py:430-433)
            except Exception as e:                      # the reference carries on with a NaN plane (dataloaders.py:437-440)
                print(f"Error opening {fn}: {e}")
        planes.append(img)
    return np.stack([np.full(shape, np.nan) if p is None else p for p in planes]), pix_to_radec


def generate_overlap_coords(img_shape, cutout_size, overlap):
    """utils/dataloaders.py:478-505: top-left corners of a regular grid of windows with the given overlap, plus windows
    flush with the bottom / right edges when the step does not divide the tile."""
    H, W = img_shape
    step = int(cutout_size * (1 - overlap))
    rows, cols = range(0, H - cutout_size + 1, step), range(0, W - cutout_size + 1, step)
    coords = [(i, j) for i in rows for j in cols]
    if H % step != 0:
        coords += [(H - cutout_size, j) for j in cols]
    if W % step != 0:
        coords += [(i, W - cutout_size) for i in rows]
    if H % step != 0 and W % step != 0:
        coords.append((H - cutout_size, W - cutout_size))
    return coords


def random_cutouts(input_array, img_size, n_cutouts, pix_to_radec=None):
    """utils/dataloaders.py:449-476 (host arrays; numpy's global generator draws the corners, rows first)."""
    C, H, W = input_array.shape
    hs = np.random.randint(0, H - img_size + 1, size=n_cutouts)
    ws = np.random.randint(0, W - img_size + 1, size=n_cutouts)
    out = np.stack([input_array[:, h:h + img_size, w:w + img_size] for h, w in zip(hs, ws)])
    if pix_to_radec is None:
        return out
    ra, dec = pix_to_radec(hs + img_size // 2, ws + img_size // 2)
    return out, np.vstack((ra, dec)).T


def overlapping_cutouts(input_array, img_size, overlap, pix_to_radec=None):
    """utils/dataloaders.py:507-536."""
    coords = generate_overlap_coords(input_array.shape[1:], img_size, overlap)
    out = np.stack([input_array[:, h:h + img_size, w:w + img_size] for h, w in coords])
    if pix_to_radec is None:
        return out
    ra, dec = pix_to_radec([h + img_size // 2 for h, _ in coords], [w + img_size // 2 for _, w in coords])
    return out, np.vstack((ra, dec)).T


class FitsDataset(torch.utils.data.Dataset):
    """utils/dataloaders.py:538-654: item = one sky patch -> (cutouts [M, batch, C, S, S], masks, ra_dec [M, batch, 2]) with
    M = cutouts_per_tile // batch_size.  Same constructor, same draws (numpy's global generator for the window corners, rows
    first; the reference's MaskGenerator semantics for the masks), same clip; the work is laid out for the GPU:

    * every band file is memory-mapped and its pixel bytes are copied to HBM as they are (big-endian floats, 70 MB per
      4k x 4k band; a nine-band tile is 0.6 GB of 288); missing or unreadable bands are NaN planes;
    * ONE launch (``skyemb_tile_cutouts``) decodes, cuts and clips all windows of the tile; the per-channel patch masks come
      from the device generator (``skyemb_simmim_mask_from_noise``);
    * RA / Dec of the window centres: ``fits_lite.TanSipWCS`` on the host (a few thousand points), with the reference's
      argument order (row index as x, column index as y: dataloaders.py:430-433, 472).

    ``transform`` (a per-tensor callable) is applied to the whole [n, C, S, S] stack like the reference does.  The returned
    tensors live on ``device``; DataLoader workers are not used (num_workers is accepted and ignored)."""

    def __init__(self, fits_paths, patch_size=8, max_mask_ratio=None, bands=('G', 'R', 'I', 'Z', 'Y'), min_bands=5, img_size=64,
                 cutouts_per_tile=1024, batch_size=64, ra_dec=False, transform=None, pixel_min=-3., pixel_max=None,
                 use_calexp=True, use_overlap=False, overlap=0.5, device="cuda"):
        self.fits_paths, self.bands = fits_paths, list(bands)
        self.img_size, self.patch_size, self.cutouts_per_tile, self.batch_size = img_size, patch_size, cutouts_per_tile, batch_size
        self.ra_dec, self.transform, self.pixel_min, self.pixel_max = ra_dec, transform, pixel_min, pixel_max
        self.use_calexp, self.use_overlap, self.overlap = use_calexp, use_overlap, overlap
        self.max_mask_ratio = max_mask_ratio
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.band_filenames = find_HSC_bands(fits_paths, self.bands, min_bands, use_calexp=use_calexp)
        self.mask_generator = None if max_mask_ratio is None else MaskGenerator(
            input_size=img_size, patch_size=patch_size, max_mask_ratio=max_mask_ratio, num_mask_chans=len(self.bands))

    def __len__(self):
        return len(self.band_filenames)

    def _tile_to_device(self, filenames):
        """-> (tile uint32-view tensor [C, H, W] on the device, big_endian int32 [C], header of the first readable band)."""
        from .. import fits_lite
        hdus = []
        for fn in filenames:
            hdu = None
            if fn != "None":
                try:
                    hdu = fits_lite.read_image_hdu(fn, 1)
                except Exception as e:
                    print(f"Error opening {fn}: {e}")
            hdus.append(hdu)
        ref = next((h for h in hdus if h is not None), None)
        if ref is None:
            raise RuntimeError(f"no readable band among {filenames}")
        H, W = ref.shape
        tile = torch.empty(len(hdus), H, W, dtype=torch.float32, device=self.device)
        be = torch.zeros(len(hdus), dtype=torch.int32)
        for c, h in enumerate(hdus):
            if h is None or h.shape != (H, W):
                tile[c].fill_(float("nan"))
            elif h.bitpix == -32 and h.bscale == 1.0 and h.bzero == 0.0:
                # the file's bytes as they are (the kernel swaps the byte order): native threads copy the mapped band into a
                # pinned buffer (pageable -> device copies run at ~4 GB/s, this path at the host's memcpy rate), two buffers
                # in turn so that the copy of one band overlaps the H2D of the previous one
                slot = self._pin_slot = 1 - getattr(self, "_pin_slot", 1)
                pins = self.__dict__.setdefault("_pins", [None, None])
                evs = self.__dict__.setdefault("_pin_events", [None, None])
                if pins[slot] is None or pins[slot].numel() != H * W:
                    pins[slot] = torch.empty(H * W, dtype=torch.int32).pin_memory()
                if evs[slot] is not None:
                    evs[slot].synchronize()                    # the H2D that last read this buffer has finished
                rows = getattr(self, "_row_idx", None)
                if rows is None or len(rows) != H:
                    rows = self._row_idx = np.arange(H, dtype=np.int64)
                raw = np.asarray(h.raw)
                from .._lib import check, lib
                check(lib().skyemb_gather_rows_host(raw.ctypes.data, W * 4, rows.ctypes.data, H, H, pins[slot].data_ptr(), 16),
                      "skyemb_gather_rows_host")
                tile[c].view(torch.int32).view(-1).copy_(pins[slot], non_blocking=True)
                evs[slot] = torch.cuda.Event()
                evs[slot].record(torch.cuda.current_stream(self.device))
                be[c] = 1
            else:                                      # integer / double / scaled images: decoded on the host
                tile[c].copy_(torch.from_numpy(np.ascontiguousarray(h.array(), dtype=np.float32)))
        return tile, be.to(self.device), ref.header

    def __getitem__(self, idx):
        from .. import fits_lite, ops
        S, B = self.img_size, self.batch_size
        tile, be, header = self._tile_to_device(self.band_filenames[idx])
        C, H, W = tile.shape
        if self.use_overlap:
            coords = generate_overlap_coords((H, W), S, self.overlap)
            hs, ws = np.array([c[0] for c in coords]), np.array([c[1] for c in coords])
        else:
            draw = getattr(self, "rng", None) or np.random      # a loader hands over its own generator (see _TileLoader)
            hs = draw.randint(0, H - S + 1, size=self.cutouts_per_tile)
            ws = draw.randint(0, W - S + 1, size=self.cutouts_per_tile)
        n = len(hs)
        cutouts = torch.empty(n, C, S, S, device=self.device)
        ops.tile_cutouts(tile, be, torch.from_numpy(hs.astype(np.int32)).to(self.device), torch.from_numpy(ws.astype(np.int32)).to(self.device),
                         S, cutouts, lo=self.pixel_min, hi=self.pixel_max)
        if self.transform is not None:
            cutouts = self.transform(cutouts)
        M = n // B
        out = [cutouts[:M * B].reshape(M, B, C, S, S)]
        if self.mask_generator is not None:
            grid = S // self.patch_size
            masks = torch.empty(M * B, C, S, S, device=self.device)
            ops.simmim_mask_from_noise(torch.rand(M * B, C, grid * grid, device=self.device), torch.rand(M * B, device=self.device),
                                       self.max_mask_ratio, grid, self.patch_size, masks)
            out.append(masks.reshape(M, B, C, S, S))
        else:
            out.append(torch.zeros(M, B, device=self.device))
        if self.ra_dec:
            wcs = fits_lite.TanSipWCS(header)
            ra, dec = wcs.all_pix2world(hs + S // 2, ws + S // 2, 0)      # (row, column) as (x, y), like the reference
            rd = torch.from_numpy(np.vstack((ra, dec)).T.astype(np.float32))[:M * B].reshape(M, B, 2)
            out.append(rd.to(self.device))
        return tuple(out)


class _TileLoader:
    """What ``DataLoader(dataset, batch_size=1, shuffle=...)`` yields for a FitsDataset -- every tensor with a leading
    dimension of 1 (the reference's loop indexes it away: pretrain_mim.py:143-150) -- without worker processes: the items
    are device tensors.  With ``prefetch`` the NEXT tile is read, uploaded and cut on a side stream by a background thread
    while the caller trains on the current one (a nine-band 4k x 4k tile is 0.6 GB of host reads and H2D: tens of
    milliseconds, about as long as the training steps its cutouts feed)."""

    def __init__(self, dataset, shuffle, prefetch=True):
        self.dataset, self.shuffle, self.prefetch = dataset, shuffle, prefetch
        self.batch_size, self.num_workers = 1, 0

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        # One private generator per pass, seeded in the CALLER's thread from numpy's global state (the reference seeds
        # that one): window corners are then drawn from it, never from the global generator inside the producer thread,
        # so a run is reproducible whatever else draws from numpy meanwhile -- and prefetching does not change the draws.
        rng = np.random.RandomState(np.random.randint(0, 2 ** 31 - 1))
        order = rng.permutation(len(self.dataset)) if self.shuffle else np.arange(len(self.dataset))
        self.dataset.rng = rng
        if not self.prefetch or len(order) < 2 or self.dataset.device.type != "cuda":
            try:
                for i in order:
                    yield tuple(t.unsqueeze(0) for t in self.dataset[int(i)])
            finally:
                self.dataset.rng = None
            return
        import queue
        import threading
        dev = self.dataset.device
        side = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=1)                      # one tile ahead: two tiles resident
        stop = threading.Event()

        def put(x):
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                torch.cuda.set_device(dev)
                for i in order:
                    if stop.is_set():
                        return
                    with torch.cuda.stream(side):
                        item = self.dataset[int(i)]
                        done = torch.cuda.Event()
                        done.record(side)
                    if not put((item, done)):
                        return
                put(None)
            except BaseException as e:                  # surfaces in the consumer
                put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    return
                if isinstance(got, BaseException):
                    raise got
                item, done = got
                main = torch.cuda.current_stream(dev)
                main.wait_event(done)
                for t in item:
                    t.record_stream(main)                   # allocated on the side stream, consumed on the caller's
                yield tuple(t.unsqueeze(0) for t in item)
        finally:
            # the consumer is done or has abandoned the generator (iteration budget reached, an exception): stop the
            # producer, free the tile it may be holding and wait for it
            stop.set()
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=30.0)
            self.dataset.rng = None


def build_fits_dataloader(fits_paths, bands, min_bands, batch_size, num_workers, patch_size=8, max_mask_ratio=None, img_size=64,
                          cutouts_per_tile=1024, use_calexp=True, augment=False, brightness=0.8, noise=0.01, nan_channels=2,
                          shuffle=True, ra_dec=True, transforms=None, use_overlap=False, overlap=0.5, device="cuda", prefetch=True):
    """utils/dataloaders.py:108-132 (+ ``device`` / ``prefetch``: see _TileLoader)."""
    if transforms is None and augment:
        # (an Augmenter maps a [n, C, S, S] stack to one augmented copy of every cutout: one launch for the whole tile)
        transforms = get_augmentations(img_size=img_size, flip=True, crop=True, brightness=brightness, noise=noise, nan_channels=nan_channels)
    dataset = FitsDataset(fits_paths, patch_size=patch_size, max_mask_ratio=max_mask_ratio, bands=bands, min_bands=min_bands,
                          img_size=img_size, cutouts_per_tile=cutouts_per_tile, batch_size=batch_size, ra_dec=ra_dec,
                          transform=transforms, use_calexp=use_calexp, use_overlap=use_overlap, overlap=overlap, device=device)
    return _TileLoader(dataset, shuffle, prefetch)
