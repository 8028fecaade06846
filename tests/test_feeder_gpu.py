"""GPU: the batched HDF5 -> HBM feeder (sky_embeddings_amd.feeder) against the per-item dataset mirror
(utils/dataloaders.py:285-328 semantics: clip at pixel_min, NaN kept, centre crop, RA/Dec)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(tmp_path, n=300, size=72, nan_fraction=0.05, chunked=False):
    from sky_embeddings_amd import hdf5_lite
    rng = np.random.default_rng(3)
    cut = (rng.standard_normal((n, 5, size, size), dtype=np.float32) * 3).astype(np.float32)   # values below -3 exist
    cut[rng.random((n, 5)) < nan_fraction] = np.nan
    path = str(tmp_path / "cutouts.h5")
    chunks = None
    if chunked:     # what the reference's ETL writes: resizable datasets with h5py's automatic chunk shapes
        chunks = {"cutouts": hdf5_lite.h5py_guess_chunk((0, 5, size, size), 4), "ra": (1024,), "dec": (1024,)}
    hdf5_lite.write_datasets(path, {"cutouts": cut, "ra": rng.uniform(0, 360, n).astype(np.float32),
                                    "dec": rng.uniform(-90, 90, n).astype(np.float32)}, chunks=chunks)
    return path, n


@pytest.mark.parametrize("chunked", [False, True])
def test_feeder_matches_per_item_dataset(tmp_path, chunked):
    from sky_embeddings_amd.feeder import CutoutFeeder
    from sky_embeddings_amd.utils.dataloaders import H5Dataset
    path, n = _make(tmp_path, chunked=chunked)
    ds = H5Dataset(path, img_size=64, patch_size=16, num_channels=5, max_mask_ratio=None)
    fd = CutoutFeeder(path, batch_size=32, img_size=64, shuffle=False, drop_last=False, depth=2, threads=3, epochs=2)
    assert len(fd) == (n + 31) // 32
    seen = 0
    for cut, mask, radec in fd:
        cut, radec = cut.cpu(), radec.cpu()          # copy out before the slot is recycled
        for j in range(cut.shape[0]):
            ref_c, ref_m, ref_r = ds[(seen + j) % n]
            assert torch.equal(torch.nan_to_num(cut[j], nan=-77.0), torch.nan_to_num(ref_c, nan=-77.0))
            assert torch.equal(radec[j], ref_r)
        assert float(mask.abs().sum()) == 0.0 and mask.shape == cut.shape
        seen += cut.shape[0]
    assert seen == 2 * n


def test_feeder_shards_are_disjoint_and_shuffled(tmp_path):
    from sky_embeddings_amd.feeder import CutoutFeeder
    path, n = _make(tmp_path, n=256, size=64, nan_fraction=0.0)
    ras = []
    for rank in range(2):
        fd = CutoutFeeder(path, batch_size=16, img_size=64, shuffle=True, seed=5, rank=rank, world_size=2)
        ras.append(torch.cat([r[:, 0].cpu().clone() for _, _, r in fd]))
    both = torch.cat(ras)
    assert len(both) == n and len(torch.unique(both)) == n           # every cutout exactly once across the ranks
    assert not torch.equal(ras[0], torch.sort(ras[0]).values)        # shuffled
    fd = CutoutFeeder(path, batch_size=16, img_size=64, shuffle=True, seed=5, rank=0, world_size=2)
    again = torch.cat([r[:, 0].cpu().clone() for _, _, r in fd])
    assert torch.equal(again, ras[0])                                # same seed, same order


def _make_tiles(root, n_patches=2, H=200, W=232, bands=("G", "R", "I", "Z", "Y"), missing=((0, "Z"),), int_band=None, gz_band=None):
    """Synthetic survey tiles in the reference's layout (one FITS file per band and patch, image in HDU 1, TAN-SIP header)."""
    from sky_embeddings_amd import fits_lite
    rng = np.random.default_rng(7)
    hdr = {"CTYPE1": "RA---TAN-SIP", "CTYPE2": "DEC--TAN-SIP", "CRPIX1": 110.5, "CRPIX2": 98.0, "CRVAL1": 35.3, "CRVAL2": -4.1,
           "CD1_1": -4.66e-5, "CD1_2": 2.0e-7, "CD2_1": 1.0e-7, "CD2_2": 4.66e-5, "A_ORDER": 2, "B_ORDER": 2, "A_0_2": 2e-7, "B_2_0": -1e-7}
    tiles = {}
    for k in range(n_patches):
        patch = f"98{13 + k}-4,{k}"
        planes = []
        for b in bands:
            img = (rng.standard_normal((H, W)) * 3).astype(np.float32)       # values below -3 exist (clip)
            img[rng.random((H, W)) < 0.01] = np.nan
            if (k, b) in missing:
                planes.append(np.full((H, W), np.nan, np.float32))
                continue
            if int_band == b:           # an integer image with BSCALE / BZERO: decoded on the host
                q = np.round(np.nan_to_num(img) * 100).astype(np.int16)
                fits_lite.write_image_fits(os.path.join(root, f"calexp-HSC-{b}-{patch}.fits"), q, dict(hdr, BSCALE=0.01, BZERO=0.0), bitpix=16)
                planes.append((q * 0.01).astype(np.float32))
            elif gz_band == b:         # a tile-compressed image (lossless gzip, byte-shuffled): decoded on the host
                fits_lite.write_compressed_image_fits(os.path.join(root, f"calexp-HSC-{b}-{patch}.fits"), img, dict(hdr, CRVAL1=35.3 + k), "GZIP_2", 1)
                planes.append(img)
            else:
                fits_lite.write_image_fits(os.path.join(root, f"calexp-HSC-{b}-{patch}.fits"), img, dict(hdr, CRVAL1=35.3 + k))
                planes.append(img)
        tiles[patch] = np.stack(planes)
    return tiles, hdr


def test_tile_sampler_matches_oracle(tmp_path):
    """FitsDataset (utils/dataloaders.py:538-654) with the tile resident in HBM: windows, clip, missing band, an integer band, a
    gzip-compressed band,
    batch layout and the RA / Dec of the window centres against the CPU restatement for the same numpy draws."""
    import os as _os
    from oracle import tile_oracle as to
    from sky_embeddings_amd.utils.dataloaders import FitsDataset, build_fits_dataloader, load_fits_bands
    tiles, hdr = _make_tiles(str(tmp_path), int_band="Y", gz_band="G")      # (G is also the band the world coordinates come from)
    ds = FitsDataset([str(tmp_path)], patch_size=8, max_mask_ratio=0.9, bands=["G", "R", "I", "Z", "Y"], min_bands=4, img_size=64,
                     cutouts_per_tile=70, batch_size=16, ra_dec=True)
    assert len(ds) == 2
    for idx in range(2):
        names = ds.band_filenames[idx]
        patch = "-".join(_os.path.basename(next(n for n in names if n != "None")).split("-")[-2:])[:-5]
        tile = tiles[patch]
        host, pix_to_radec = load_fits_bands(names, return_wc=True)                     # the host path of the mirror
        assert np.array_equal(np.nan_to_num(host.astype(np.float32), nan=-77), np.nan_to_num(tile, nan=-77))
        np.random.seed(100 + idx)
        cut, masks, rd = ds[idx]
        np.random.seed(100 + idx)
        hs = np.random.randint(0, tile.shape[1] - 64 + 1, size=70)
        ws = np.random.randint(0, tile.shape[2] - 64 + 1, size=70)
        ref = to.cutouts_np(tile, hs, ws, 64, pixel_min=-3.0)[:64].reshape(4, 16, 5, 64, 64)
        got = cut.cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(np.nan_to_num(got, nan=-77), np.nan_to_num(ref, nan=-77))
        assert float(np.nanmin(got)) >= -3.0 and np.isnan(got).any()
        h2 = dict(hdr, CRVAL1=35.3 + int(patch.split(",")[1]))
        ra, dec = to.tan_sip_pix2world(h2, hs + 32, ws + 32, 0)                          # (row, column) as (x, y), like the reference
        assert np.allclose(rd.cpu().numpy().reshape(-1, 2), np.vstack((ra, dec)).T[:64].astype(np.float32), rtol=0, atol=1e-5)
        m = masks.cpu()
        assert m.shape == (4, 16, 5, 64, 64) and bool(((m == 0) | (m == 1)).all())
        per_channel = m[:, :, :, ::8, ::8].sum(dim=(3, 4))
        assert bool((per_channel == per_channel[:, :, :1]).all())                        # same count in every channel of a sample
    loader = build_fits_dataloader([str(tmp_path)], ["G", "R", "I", "Z", "Y"], 4, batch_size=16, num_workers=3, patch_size=8,
                                   max_mask_ratio=None, img_size=64, cutouts_per_tile=40, shuffle=False, ra_dec=True)
    np.random.seed(5)
    items = [tuple(t.clone() for t in it) for it in loader]                               # next tile prefetched on a side stream
    assert len(items) == 2 and items[0][0].shape == (1, 2, 16, 5, 64, 64) and items[0][1].shape == (1, 2, 16) and items[0][2].shape == (1, 2, 16, 2)
    loader.prefetch = False
    np.random.seed(5)
    for a, b in zip(items, loader):                                                        # same draws, same tensors without it
        assert torch.equal(torch.nan_to_num(a[0], nan=-77.0), torch.nan_to_num(b[0], nan=-77.0)) and torch.equal(a[2], b[2])
    over = FitsDataset([str(tmp_path)], bands=["G", "R", "I", "Z", "Y"], min_bands=4, img_size=64, batch_size=8, use_overlap=True, overlap=0.5)
    cut, masks = over[0]
    from sky_embeddings_amd.utils.dataloaders import generate_overlap_coords
    coords = generate_overlap_coords((200, 232), 64, 0.5)
    assert cut.shape == (len(coords) // 8, 8, 5, 64, 64)
