"""GPU: the batched HDF5 -> HBM feeder (sky_embeddings_amd.feeder) against the per-item dataset mirror
(utils/dataloaders.py:285-328 semantics: clip at pixel_min, NaN kept, centre crop, RA/Dec)."""
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
