"""utils/misc.py mirror (what the hot path and its validation hooks use: SURVEY.md §2 row 8)."""
import argparse

import numpy as np


def str2bool(v):
    """utils/misc.py:6-7."""
    return str(v).lower() in ("yes", "true", "t", "1")


def parseArguments():
    """utils/misc.py:9-33: ``model_name [-v N] [-ct MIN] [-dd DIR]``."""
    parser = argparse.ArgumentParser('Training for Masked Image Modelling', add_help=False)
    parser.add_argument("model_name", help="Name of model.", type=str)
    parser.add_argument("-v", "--verbose_iters",
                        help="Number of batch  iters after which to evaluate val set and display output.",
                        type=int, default=10000)
    parser.add_argument("-ct", "--cp_time", help="Number of minutes after which to save a checkpoint.",
                        type=float, default=15)
    parser.add_argument("-dd", "--data_dir", help="Data directory if different from sky_embeddings/data/",
                        type=str, default=None)
    return parser


def calculate_snr(images, n_central_pix):
    """utils/misc.py:138-163: mean of the central n x n region over the std of the rest, per channel."""
    batch_size, n_channels, img_size, _ = images.shape
    start = (img_size - n_central_pix) // 2
    end = start + n_central_pix
    central = images[:, :, start:end, start:end]
    mask = np.ones((img_size, img_size), dtype=bool)
    mask[start:end, start:end] = False
    surround = images[:, :, mask].reshape(batch_size, n_channels, -1)
    return np.mean(central, axis=(2, 3)) / (np.std(surround, axis=2) + 1e-8)


def h5_snr(h5_path, n_central_pix=8, batch_size=5000, num_samples=None):
    """utils/misc.py:165-180, reading through this package's HDF5 access layer."""
    from .dataloaders import open_h5
    snr_vals = []
    with open_h5(h5_path) as f:
        cut = f['cutouts']
        if num_samples is None:
            num_samples = len(cut)
        for i in range(0, num_samples, batch_size):
            end = min(num_samples, i + batch_size)
            snr_vals.append(calculate_snr(np.asarray(cut[i:end]), n_central_pix))
    return np.concatenate(snr_vals)


def central_indices(grid, n):
    """utils/misc.py:68-97: (row, col) index pairs of the central sqrt(n) x sqrt(n) block of a 2-D grid, row-major."""
    side = int(n ** 0.5)
    if side * side != n:
        raise ValueError("n must be a perfect square to form a square patch of pixels.")
    rows = np.arange(side) + grid.shape[0] // 2 - side // 2
    cols = np.arange(side) + grid.shape[1] // 2 - side // 2
    return np.stack(np.meshgrid(rows, cols, indexing="ij"), axis=-1).reshape(-1, 2)


def select_centre(latent, n_patches):
    """utils/misc.py:99-117: the central ``n_patches`` patch tokens of [b, L, features] (L a square number, raster order)."""
    side = int(latent.shape[1] ** 0.5)
    ij = central_indices(np.empty((side, side)), n_patches)
    return latent[:, ij[:, 0] * side + ij[:, 1]]


def calculate_n_samples_per_class(class_counts, num_train, balanced=False):
    """utils/misc.py:34-46: how many samples of each class a training subset of ``num_train`` takes -- the same number of every
    class (balanced: limited by the rarest class), or each class's share of the whole set (rounded down)."""
    if balanced:
        n = min(num_train // len(class_counts), min(class_counts.values()))
        return {c: n for c in class_counts}
    total = sum(class_counts.values())
    return {c: int((cnt / total) * num_train) for c, cnt in class_counts.items()}


def select_training_indices(data_file_path, num_train, balanced=False):
    """utils/misc.py:48-66: indices of the FIRST n samples of every class of the file's ``class`` column."""
    from ..hdf5_lite import File
    with File(data_file_path, 'r') as f:
        classes = np.asarray(f['class'])
    values, counts = np.unique(classes, return_counts=True)
    per_class = calculate_n_samples_per_class(dict(zip(values.tolist(), counts.tolist())), num_train, balanced)
    out = []
    for c, n in per_class.items():
        out.extend(np.where(classes == c)[0][:n].tolist())
    return out
