"""Third-party-written file fixtures that pin ``hdf5_lite`` and ``fits_lite`` (tests/test_io_fixtures_cpu.py).

The image's system python has neither h5py nor astropy, but ``/opt/conda`` does (python 3.9, h5py 3.3.0 / HDF5 1.10.6,
astropy 4.3.1).  Run THIS script with that interpreter and a clean environment:

    env -i PATH=/opt/conda/bin:/usr/bin:/bin /opt/conda/bin/python3.9 tests/golden/make_io_fixtures.py

It writes, under tests/golden/io/ (data only; ``--out DIR`` writes elsewhere, ``--full`` uses the reference's real
5 x 64 x 64 cutout geometry for the chunked file -- 10 MB, so only the live test makes it, in a temporary directory):

* ``h5py_contiguous.h5``  -- the layout ``data_processing/utils.py:346-361`` (``cutouts_to_hdf5``) writes: fixed-shape
  ``create_dataset(name, shape, dtype='f')`` for cutouts [n, C, S, S] and the per-object columns.
* ``h5py_resizable.h5``   -- the layout ``data_processing/combine_h5.py:30-32`` / ``2_create_h5_files.py:72-74`` write:
  ``maxshape=(None, ...)`` datasets, which h5py auto-chunks, grown by ``resize`` and filled in two appends (so the
  chunk B-tree has entries allocated in non-monotonic file order), plus an integer class column.
* ``h5py_filtered.h5``    -- chunked datasets behind h5py's built-in filters (gzip, shuffle, fletcher32): ``write_h5_filtered``.
* ``astropy_f4.fits``      -- empty primary HDU + a ``>f4`` IMAGE extension with NaNs and a TAN-SIP header (what HSC
  ``calexp`` patches carry; ``utils/dataloaders.py:418-432`` reads ``hdul[1].data`` and ``WCS(hdul[1].header)``).
* ``astropy_i2_scaled.fits`` -- an int16 IMAGE extension with BSCALE / BZERO.
* ``astropy_rice_*.fits``, ``astropy_gzip*_f4_*.fits`` -- tile-compressed images (``CompImageHDU``): see ``write_compressed_fits``.
* ``io_expected.npz``      -- what h5py / astropy themselves read back from those files: the arrays, ``.data`` of both
  images, and ``WCS.all_pix2world(x, y, 0)`` / ``(x, y, 1)`` at a grid of pixels.
"""
import os
import sys

import numpy as np

for _name, _fn in (("asscalar", lambda a: a.item()), ("alen", len)):      # astropy 4.3 touches these on numpy >= 1.23
    if not hasattr(np, _name):
        setattr(np, _name, _fn)

import h5py                                   # noqa: E402
from astropy.io import fits                   # noqa: E402
from astropy.wcs import WCS                   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "io")


def cutouts(rng, n, c=5, s=64):
    a = rng.standard_normal((n, c, s, s)).astype(np.float32)
    a[rng.random(a.shape) < 0.002] = np.nan
    return a


def write_h5(expected, full):
    rng = np.random.default_rng(20260104)
    # (i) fixed-shape datasets: contiguous storage
    n = 6 if full else 2
    S = 64 if full else 8            # committed fixtures: 8 x 8 cutouts (a chunked 64 x 64 file is 10 MB: edge chunks are stored whole)
    cut = cutouts(rng, n)
    cols = {k: rng.random(n).astype(np.float32) * s for k, s in (("ra", 360.0), ("dec", 90.0), ("zspec", 2.0), ("zspec_err", 0.01))}
    p = os.path.join(OUT, "h5py_contiguous.h5")
    with h5py.File(p, "w") as f:
        d = f.create_dataset("cutouts", (n, 5, 64, 64), dtype="f")
        for i in range(n):
            d[i] = cut[i]
        for k, v in cols.items():
            dd = f.create_dataset(k, (n,), dtype="f")
            dd[:] = v
    with h5py.File(p, "r") as f:
        assert f["cutouts"].chunks is None
        for k in f.keys():
            expected["contig/" + k] = f[k][:]
    # (ii) resizable datasets: chunked storage, grown by two appends
    n1, n2 = (5, 4) if full else (150, 110)     # small geometry: several chunks along the growing axis too
    cut = cutouts(rng, n1 + n2, s=S)
    ra = rng.random(n1 + n2).astype(np.float32) * 360
    dec = (rng.random(n1 + n2).astype(np.float32) - 0.5) * 180
    cls = rng.integers(0, 3, n1 + n2).astype(np.int64)
    p = os.path.join(OUT, "h5py_resizable.h5")
    with h5py.File(p, "w") as f:
        f.create_dataset("cutouts", (0, 5, S, S), maxshape=(None, 5, S, S), dtype="f")
        f.create_dataset("ra", (0,), maxshape=(None,), dtype="f")
        f.create_dataset("dec", (0,), maxshape=(None,), dtype="f")
        f.create_dataset("class", (0,), maxshape=(None,), dtype="i8")
        lo = 0
        for m in (n1, n2):
            for k, v in (("cutouts", cut), ("ra", ra), ("dec", dec), ("class", cls)):
                f[k].resize(lo + m, axis=0)
                f[k][lo:lo + m] = v[lo:lo + m]
            lo += m
    with h5py.File(p, "r") as f:
        expected["resizable/chunks"] = np.array(f["cutouts"].chunks)
        for k in f.keys():
            expected["resizable/" + k] = f[k][:]


def write_h5_filtered(expected):
    """Chunked datasets behind h5py's built-in filters: gzip (deflate) with and without byte shuffle, fletcher32 checksums, an
    explicit chunk shape with ragged edges, a 1-D integer column, and a dataset some of whose chunks were never written."""
    rng = np.random.default_rng(20260107)
    p = os.path.join(OUT, "h5py_filtered.h5")
    cut = cutouts(rng, 37, s=8)
    cut[:, :, :3] = np.round(cut[:, :, :3], 1)                       # (compressible rows)
    with h5py.File(p, "w") as f:
        f.create_dataset("cutouts", data=cut, chunks=(8, 2, 8, 5), compression="gzip", compression_opts=4, shuffle=True)
        f.create_dataset("gz_only", data=cut[:9], chunks=(4, 5, 8, 8), compression="gzip")
        f.create_dataset("checked", data=cut[:6], chunks=(3, 5, 8, 8), fletcher32=True)
        f.create_dataset("all3", data=cut[:10], chunks=(5, 5, 4, 8), compression="gzip", shuffle=True, fletcher32=True)
        f.create_dataset("class", data=rng.integers(0, 3, 37).astype(np.int64), chunks=(16,), compression="gzip", shuffle=True)
        d = f.create_dataset("sparse", (20, 5, 8, 8), dtype="f", chunks=(4, 5, 8, 8), compression="gzip")
        d[8:12] = cut[:4]
    with h5py.File(p, "r") as f:
        for k in f.keys():
            expected["filtered/" + k] = f[k][:]
        expected["filtered/cutouts_chunks"] = np.array(f["cutouts"].chunks)


SIP = {"A_ORDER": 3, "B_ORDER": 3, "A_2_0": 2.1e-7, "A_1_1": -3.4e-7, "A_0_2": 1.2e-7, "A_3_0": 4.0e-11, "A_1_2": -2.5e-11,
       "B_2_0": -1.7e-7, "B_1_1": 2.9e-7, "B_0_2": -0.8e-7, "B_0_3": 3.1e-11, "B_2_1": 1.9e-11}


def tan_sip_header(h, w):
    hd = fits.Header()
    hd["CTYPE1"], hd["CTYPE2"] = "RA---TAN-SIP", "DEC--TAN-SIP"
    hd["CRPIX1"], hd["CRPIX2"] = 0.5 * w + 3.25, 0.5 * h - 1.5
    hd["CRVAL1"], hd["CRVAL2"] = 150.1163, 2.2057
    # HSC pixel scale 0.168"/px with a small rotation
    s, th = 0.168 / 3600.0, np.deg2rad(1.3)
    hd["CD1_1"], hd["CD1_2"] = -s * np.cos(th), s * np.sin(th)
    hd["CD2_1"], hd["CD2_2"] = s * np.sin(th), s * np.cos(th)
    for k, v in SIP.items():
        hd[k] = v
    return hd


def write_fits(expected):
    rng = np.random.default_rng(20260105)
    h, w = 96, 120
    img = rng.standard_normal((h, w)).astype(np.float32)
    img[rng.random(img.shape) < 0.01] = np.nan
    hd = tan_sip_header(h, w)
    p = os.path.join(OUT, "astropy_f4.fits")
    fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=hd)]).writeto(p, overwrite=True)
    with fits.open(p, mode="readonly", ignore_missing_simple=True) as hdul:     # the reference's call (dataloaders.py:418)
        expected["fits_f4/data"] = np.array(hdul[1].data)
        assert hdul[1].data.dtype == np.dtype(">f4")
        wcs = WCS(hdul[1].header)
        yy, xx = np.meshgrid(np.linspace(0, h - 1, 7), np.linspace(0, w - 1, 9), indexing="ij")
        x, y = xx.ravel(), yy.ravel()
        expected["fits_f4/pix_x"], expected["fits_f4/pix_y"] = x, y
        for origin in (0, 1):
            ra, dec = wcs.all_pix2world(x, y, origin)
            expected[f"fits_f4/ra_o{origin}"], expected[f"fits_f4/dec_o{origin}"] = np.asarray(ra), np.asarray(dec)
    # plain TAN (no SIP) header on the same file's geometry: the PC / CDELT form
    hd2 = fits.Header()
    hd2["CTYPE1"], hd2["CTYPE2"] = "RA---TAN", "DEC--TAN"
    hd2["CRPIX1"], hd2["CRPIX2"] = 40.0, 50.0
    hd2["CRVAL1"], hd2["CRVAL2"] = 359.98, -45.3            # RA wraps through 0 across the image
    hd2["CDELT1"], hd2["CDELT2"] = -0.168 / 3600.0 * 20, 0.168 / 3600.0 * 20
    hd2["PC1_1"], hd2["PC1_2"], hd2["PC2_1"], hd2["PC2_2"] = 0.9993908, -0.0348995, 0.0348995, 0.9993908
    raw = rng.integers(-2000, 2000, (h, w)).astype(np.int16)
    hdu = fits.ImageHDU(data=raw, header=hd2)
    hdu.scale("int16", bscale=0.25, bzero=100.0)       # stored = (physical - bzero) / bscale, header gains BSCALE / BZERO
    p = os.path.join(OUT, "astropy_i2_scaled.fits")
    fits.HDUList([fits.PrimaryHDU(), hdu]).writeto(p, overwrite=True)
    with fits.open(p, mode="readonly", ignore_missing_simple=True) as hdul:
        assert hdul[1].header["BITPIX"] == 16 and "BSCALE" in hdul[1].header
        expected["fits_i2/data"] = np.array(hdul[1].data)
        wcs = WCS(hdul[1].header)
        x = np.array([0.0, 39.0, 119.0, 60.5, 5.0])
        y = np.array([0.0, 49.0, 95.0, 10.25, 90.0])
        expected["fits_i2/pix_x"], expected["fits_i2/pix_y"] = x, y
        ra, dec = wcs.all_pix2world(x, y, 0)
        expected["fits_i2/ra_o0"], expected["fits_i2/dec_o0"] = np.asarray(ra), np.asarray(dec)


COMPRESSED = (   # name, dtype, (H, W), CompImageHDU keywords
    ("rice_i2", "int16", (37, 53), dict(compression_type="RICE_1")),
    ("rice_i4_tiles", "int32", (45, 50), dict(compression_type="RICE_1", tile_size=(16, 20))),
    ("rice_u1", "uint8", (20, 33), dict(compression_type="RICE_1")),
    ("rice_f4_nodither", "float32", (40, 64), dict(compression_type="RICE_1", quantize_level=16.0, quantize_method=-1)),
    # tiles of 100 x 100 pixels: the walk through the 10 000 random numbers restarts inside a tile; dither_seed near the table's end
    ("rice_f4_dither1", "float32", (120, 110), dict(compression_type="RICE_1", tile_size=(100, 100), quantize_level=16.0,
                                                    quantize_method=1, dither_seed=9999)),
    ("rice_f4_dither2", "float32", (48, 70), dict(compression_type="RICE_1", tile_size=(70, 12), quantize_level=8.0,
                                                  quantize_method=2, dither_seed=42)),
    ("gzip1_f4_dither1", "float32", (30, 41), dict(compression_type="GZIP_1", quantize_level=16.0, quantize_method=1, dither_seed=7)),
    ("gzip2_f4_lossless", "float32", (30, 41), dict(compression_type="GZIP_2", quantize_level=0.0)),
    ("plio_i4", "mask", (23, 37), dict(compression_type="PLIO_1")),
    ("hcompress_i2", "int16", (40, 52), dict(compression_type="HCOMPRESS_1", hcomp_scale=0, tile_size=(52, 16))),
    ("hcompress_i4_odd", "int32", (37, 45), dict(compression_type="HCOMPRESS_1", hcomp_scale=0, tile_size=(45, 13))),
    ("hcompress_f4_lossy", "float32", (48, 40), dict(compression_type="HCOMPRESS_1", hcomp_scale=2.0, quantize_level=16.0, quantize_method=1,
                                                     dither_seed=5, tile_size=(40, 16))),
    # the optional smoothing on decompression (ZNAME2 = 'SMOOTH'): lossy integers and dithered floats
    ("hcompress_i4_smooth", "smooth_int", (70, 90), dict(compression_type="HCOMPRESS_1", hcomp_scale=50, hcomp_smooth=1, tile_size=(45, 35))),
    ("hcompress_f4_smooth", "float32", (70, 90), dict(compression_type="HCOMPRESS_1", hcomp_scale=4.0, hcomp_smooth=1, quantize_level=16.0,
                                                      quantize_method=1, dither_seed=9, tile_size=(90, 32))),
)


def write_compressed_fits(expected):
    """Tile-compressed images as astropy's CompImageHDU (CFITSIO underneath) writes them, and what astropy reads back
    (``fits.open(fn)[1].data``, utils/dataloaders.py:418-421): Rice on integers of 1 / 2 / 4 bytes, Rice and gzip on
    quantised floats without dithering and with both subtractive dithers (exact zeros under SUBTRACTIVE_DITHER_2; a constant
    region, which the writer cannot quantise), 2-D tiles with ragged edges; PLIO_1 on a mask-like image; HCOMPRESS_1 lossless on
    16- and 32-bit integers (odd tile sides) and lossy (scale 2) on dithered quantised floats; HCOMPRESS_1 with smoothing on
    decompression, lossy integers (scale 50) and dithered floats (scale 4)."""
    rng = np.random.default_rng(20260106)
    for name, dtype, (h, w), kw in COMPRESSED:
        if dtype == "float32":
            yy, xx = np.mgrid[0:h, 0:w]
            img = (rng.standard_normal((h, w)) * 0.7 + 12.0 * np.exp(-((yy - h / 2) ** 2 + (xx - w / 3) ** 2) / 40.0)).astype(np.float32)
            img[rng.random(img.shape) < 0.01] = np.nan
            if kw.get("quantize_method") == 2:
                img[rng.random(img.shape) < 0.05] = 0.0
            img[: h // 3, : w // 4] = 3.5                          # constant patch
        elif dtype == "smooth_int":                                # a smooth source on noise, in hundredths
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.round((rng.standard_normal((h, w)) * 3 + 10 + 40 * np.exp(-((yy - 30) ** 2 + (xx - 50) ** 2) / 60.0)) * 100).astype(np.int32)
        elif dtype == "mask":                                      # PLIO_1: non-negative integers below 2^24, long runs
            img = rng.integers(0, 50, (h, w)).astype(np.int32)
            img[5:9] = 0
            img[10, :20] = 100000
            img[12, 3:30] = 7
            img[13] = np.arange(w) * 500
        elif dtype == "uint8":
            img = rng.integers(0, 256, (h, w)).astype(np.uint8)
            img[5:9] = 17                                          # all-zero difference blocks
        else:
            lim = 2000 if dtype == "int16" else 200000
            img = rng.integers(-lim, lim, (h, w)).astype(dtype)
            img[3:6] = 7
            img[10, 5:40] = rng.integers(np.iinfo(dtype).min, np.iinfo(dtype).max, 35).astype(dtype)   # high-entropy blocks
        hd = fits.ImageHDU(data=img, header=tan_sip_header(h, w)).header       # (astropy 4.3 wants a complete image header)
        p = os.path.join(OUT, f"astropy_{name}.fits")
        fits.HDUList([fits.PrimaryHDU(), fits.CompImageHDU(data=img, header=hd, **kw)]).writeto(p, overwrite=True)
        with fits.open(p, mode="readonly", ignore_missing_simple=True) as hdul:
            data = np.array(hdul[1].data)
            assert data.shape == (h, w)
            expected[f"fits_{name}/data"] = data
            if (dtype != "float32" and not kw.get("hcomp_scale")) or kw.get("quantize_level") == 0.0:
                assert np.array_equal(data, img, equal_nan=True)   # lossless (HCOMPRESS_1 with scale 0 included)
        with fits.open(p, mode="readonly", disable_image_compression=True) as hdul:   # the table as it is on disk
            th = hdul[1].header
            expected[f"fits_{name}/ztile"] = np.array([th["ZTILE1"], th["ZTILE2"]])
            expected[f"fits_{name}/zquantiz"] = np.array(str(th.get("ZQUANTIZ", "NONE")))
            expected[f"fits_{name}/columns"] = np.array([c.name for c in hdul[1].columns])


def main():
    global OUT
    full = "--full" in sys.argv          # the reference's real geometry (5 x 64 x 64): ~11 MB, made on the fly by the live test
    if "--out" in sys.argv:
        OUT = sys.argv[sys.argv.index("--out") + 1]
    os.makedirs(OUT, exist_ok=True)
    expected = {}
    write_h5(expected, full)
    write_h5_filtered(expected)
    write_fits(expected)
    write_compressed_fits(expected)
    expected["versions"] = np.array([f"h5py {h5py.__version__}", f"hdf5 {h5py.version.hdf5_version}",
                                     f"astropy {__import__('astropy').__version__}", f"numpy {np.__version__}",
                                     f"python {sys.version.split()[0]}"])
    np.savez_compressed(os.path.join(OUT, "io_expected.npz"), **expected)
    for fn in sorted(os.listdir(OUT)):
        print(f"{fn:28s} {os.path.getsize(os.path.join(OUT, fn)):8d} B")


if __name__ == "__main__":
    main()
