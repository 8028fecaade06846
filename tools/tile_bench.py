#!/usr/bin/env python3
"""Throughput of the survey-tile input path on full-size synthetic tiles (HSC patch geometry: 4100 x 4200 pixels per band).
usage: python tools/tile_bench.py [bands=5] [tiles=3] [cutouts_per_tile=1024] [img=64]"""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import fits_lite
from sky_embeddings_amd.utils.dataloaders import build_fits_dataloader

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cpt = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
S = int(sys.argv[4]) if len(sys.argv) > 4 else 64
bands = ["G", "R", "I", "Z", "Y", "NB0387", "NB0816", "NB0921", "NB1010"][:nb]
hdr = {"CTYPE1": "RA---TAN-SIP", "CTYPE2": "DEC--TAN-SIP", "CRPIX1": 2050.0, "CRPIX2": 2100.0, "CRVAL1": 150.0, "CRVAL2": 2.0,
       "CD1_1": -4.66e-5, "CD1_2": 0.0, "CD2_1": 0.0, "CD2_2": 4.66e-5, "A_ORDER": 2, "B_ORDER": 2, "A_2_0": 1e-8, "B_0_2": 1e-8}
root = tempfile.mkdtemp(prefix="tiles_")
rng = np.random.default_rng(0)
img = rng.standard_normal((4200, 4100)).astype(np.float32)
for k in range(nt):
    for b in bands:
        fits_lite.write_image_fits(os.path.join(root, f"calexp-HSC-{b}-9813-{k},0.fits"), img, hdr)
print(f"{nt} tiles x {nb} bands of 4200 x 4100 float32 ({nt * nb * img.nbytes / 1e9:.2f} GB) in {root}", flush=True)
for prefetch in (False, True):
    loader = build_fits_dataloader([root], bands, nb, batch_size=128, num_workers=0, patch_size=8, max_mask_ratio=0.9, img_size=S,
                                   cutouts_per_tile=cpt, shuffle=False, ra_dec=True, prefetch=prefetch)
    for _ in loader:   # warm: page cache, kernels
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for cut, msk, rd in loader:
        n += cut.shape[1] * cut.shape[2]
        # stand-in for the training steps on this tile's batches: 128 images per 37 ms (mim_19) would be cpt / 128 * 37 ms
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"prefetch={prefetch}: {n} cutouts of {nb} x {S} x {S} in {dt * 1e3:.0f} ms = {n / dt / 1e3:.1f} k cutouts/s ({dt / nt * 1e3:.0f} ms per tile, "
          f"{nt * nb * img.nbytes / dt / 1e9:.1f} GB/s of tile bytes)", flush=True)
import shutil
shutil.rmtree(root)
