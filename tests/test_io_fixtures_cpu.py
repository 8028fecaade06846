"""CPU: ``hdf5_lite`` and ``fits_lite`` against files WRITTEN BY h5py 3.3.0 / astropy 4.3.1 and against what those libraries
read back from them (tests/golden/io/, made by tests/golden/make_io_fixtures.py under /opt/conda's interpreter).

This is what pins the product's file readers to the libraries the reference uses (``utils/dataloaders.py:281-328`` h5py,
``:418-432`` astropy ``fits.open`` + ``WCS.all_pix2world``); before round 4 they were checked against their own writers only.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from sky_embeddings_amd import fits_lite, hdf5_lite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IO = os.path.join(ROOT, "tests", "golden", "io")
CONDA_PY = "/opt/conda/bin/python3.9"


def same(a, b):
    """bit-equal including NaN payload positions"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def check_h5(folder, exp, tmp_path):
    # (i) fixed-shape datasets written by h5py: contiguous layout (data_processing/utils.py:346-361)
    with hdf5_lite.File(os.path.join(folder, "h5py_contiguous.h5")) as f:
        assert sorted(f.keys()) == ["cutouts", "dec", "ra", "zspec", "zspec_err"]
        for k in f.keys():
            assert same(f[k][:], exp["contig/" + k]), k
        assert same(f["cutouts"][1], exp["contig/cutouts"][1])
    # (ii) maxshape=(None, ...) datasets, grown by two appends: h5py's automatic chunks behind a v1 B-tree
    with hdf5_lite.File(os.path.join(folder, "h5py_resizable.h5")) as f:
        assert sorted(f.keys()) == ["class", "cutouts", "dec", "ra"]
        d = f["cutouts"]
        full = exp["resizable/cutouts"]
        assert d.shape == full.shape and d.dtype == np.float32
        assert tuple(d.chunks) == tuple(int(v) for v in exp["resizable/chunks"])
        # the chunk shape h5py picked is the one our own writer predicts for this shape (feeder tests rely on that)
        assert hdf5_lite.h5py_guess_chunk((0,) + full.shape[1:], 4) == tuple(d.chunks)
        for i in (0, 1, full.shape[0] // 2, full.shape[0] - 1):
            assert same(d[i], full[i]), i
        assert same(d[:], full)
        assert same(f["class"][:], exp["resizable/class"]) and f["class"].dtype == np.int64
        assert same(f["ra"][:], exp["resizable/ra"]) and same(f["dec"][:], exp["resizable/dec"])


def check_h5_filtered(folder, exp, tmp_path):
    """gzip / shuffle / fletcher32 chunks written by h5py: indexed reads and the feeder's one-off contiguous copy, bit for bit."""
    p = str(tmp_path / "f.h5")
    shutil.copy(os.path.join(folder, "h5py_filtered.h5"), p)
    with hdf5_lite.File(p) as f:
        assert sorted(f.keys()) == ["all3", "checked", "class", "cutouts", "gz_only", "sparse"]
        assert [fid for fid, _ in f["cutouts"].filters] == [2, 1] and [fid for fid, _ in f["all3"].filters] == [2, 1, 3]
        assert tuple(f["cutouts"].chunks) == tuple(int(v) for v in exp["filtered/cutouts_chunks"]) == (8, 2, 8, 5)
        assert same(f["class"][:], exp["filtered/class"]) and same(f["class"][20:30], exp["filtered/class"][20:30])    # per-chunk reads
        for k in ("cutouts", "gz_only", "checked", "all3", "sparse"):
            want = exp["filtered/" + k]
            assert same(f[k][:], want), k                    # through the contiguous cache (Dataset._array)
            assert same(f[k][want.shape[0] - 1], want[-1]) and same(np.asarray(f[k]), want), k
        assert not exp["filtered/sparse"][:8].any() and exp["filtered/sparse"][8:12].any()
    with hdf5_lite.File(p) as f:                             # a fresh handle: the cache file is found and reused
        assert same(f["cutouts"][5], exp["filtered/cutouts"][5])
        d = f["cutouts"]
        d._mm = None
        assert same(d._read_chunked(np.array([36, 0, 17])), exp["filtered/cutouts"][[36, 0, 17]])       # without the cache


def check_fits(folder, exp):
    hdu = fits_lite.read_image_hdu(os.path.join(folder, "astropy_f4.fits"), hdu=1)
    assert hdu.bitpix == -32 and hdu.raw.dtype == np.dtype(">f4")
    want = exp["fits_f4/data"]
    assert same(hdu.array(), want.astype(np.float32))           # astropy hands out big-endian floats; values identical
    wcs = fits_lite.TanSipWCS(hdu.header)
    assert wcs.sip and len(wcs.a) == 5 and len(wcs.b) == 5
    x, y = exp["fits_f4/pix_x"], exp["fits_f4/pix_y"]
    for origin in (0, 1):
        ra, dec = wcs.all_pix2world(x, y, origin)
        # 1e-11 deg = 3.6e-8 arcsec; wcslib iterates nothing here (pix -> world is closed form), so only rounding differs
        assert np.abs(ra - exp[f"fits_f4/ra_o{origin}"]).max() < 1e-11
        assert np.abs(dec - exp[f"fits_f4/dec_o{origin}"]).max() < 1e-11
    hdu = fits_lite.read_image_hdu(os.path.join(folder, "astropy_i2_scaled.fits"), hdu=1)
    assert hdu.bitpix == 16 and hdu.bscale == 0.25 and hdu.bzero == 100.0
    want = exp["fits_i2/data"]                                  # astropy: float32 = raw * BSCALE + BZERO for 16-bit integers
    got = hdu.array()
    assert np.array_equal(got.astype(want.dtype), want)
    wcs = fits_lite.TanSipWCS(hdu.header)
    assert not wcs.sip
    ra, dec = wcs.all_pix2world(exp["fits_i2/pix_x"], exp["fits_i2/pix_y"], 0)
    dra = (ra - exp["fits_i2/ra_o0"] + 180.0) % 360.0 - 180.0  # the field straddles RA = 0
    assert np.abs(dra).max() < 1e-11 and np.abs(dec - exp["fits_i2/dec_o0"]).max() < 1e-11
    assert ra.min() >= 0.0 and ra.max() < 360.0 and (ra < 1).any() and (ra > 359).any()


COMPRESSED = ("rice_i2", "rice_i4_tiles", "rice_u1", "rice_f4_nodither", "rice_f4_dither1", "rice_f4_dither2", "gzip1_f4_dither1",
              "gzip2_f4_lossless", "plio_i4", "hcompress_i2", "hcompress_i4_odd", "hcompress_f4_lossy", "hcompress_i4_smooth",
              "hcompress_f4_smooth")


def check_compressed_fits(folder, exp):
    """Tile-compressed images written by astropy's CompImageHDU: what ``fits_lite`` decodes == what astropy's ``.data`` holds,
    bit for bit (Rice on 1 / 2 / 4-byte integers; Rice and gzip on quantised floats, undithered and with both subtractive
    dithers -- the 100 x 100 tiles of ``rice_f4_dither1`` walk past the end of the random table, its ZDITHER0 = 9999 wraps the
    seed index; unquantisable tiles in GZIP_COMPRESSED_DATA; PLIO_1; HCOMPRESS_1 lossless on integers with odd tile sides,
    lossy on dithered floats, and with its smoothing on decompression)."""
    for name in COMPRESSED:
        hdu = fits_lite.read_image_hdu(os.path.join(folder, f"astropy_{name}.fits"), hdu=1)
        want = exp[f"fits_{name}/data"]
        got = hdu.array()
        assert got.shape == want.shape, name
        if want.dtype.kind == "f":
            assert hdu.bitpix == -32 and hdu.raw.dtype == np.dtype(">f4")
            assert same(got, want.astype(np.float32)), name          # incl. the NaN pixels astropy reads back
        else:
            assert hdu.bitpix == 8 * want.dtype.itemsize
            assert np.array_equal(got.astype(want.dtype), want) and np.array_equal(got, want.astype(np.float64)), name
        assert fits_lite.TanSipWCS(hdu.header).sip                  # the image header rides along
    # the cases are what their names say (the writer's own header, read with astropy's decompression switched off)
    assert str(exp["fits_rice_f4_dither1/zquantiz"]) == "SUBTRACTIVE_DITHER_1" and tuple(exp["fits_rice_f4_dither1/ztile"]) == (100, 100)
    assert str(exp["fits_rice_f4_dither2/zquantiz"]) == "SUBTRACTIVE_DITHER_2" and tuple(exp["fits_rice_i4_tiles/ztile"]) == (16, 20)
    assert "GZIP_COMPRESSED_DATA" in list(exp["fits_rice_f4_nodither/columns"])
    assert int((exp["fits_rice_f4_dither2/data"] == 0).sum()) > 100  # exact zeros survive SUBTRACTIVE_DITHER_2
    # the smoothing does something: with the file's SMOOTH flag cleared the same tiles decode to other pixels
    src = os.path.join(folder, "astropy_hcompress_i4_smooth.fits")
    raw = bytearray(open(src, "rb").read())
    at = raw.index(b"ZVAL2   =")
    assert raw[at + 29:at + 30] == b"1"
    raw[at + 29:at + 30] = b"0"
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "ns.fits")
        open(p, "wb").write(bytes(raw))
        plain = fits_lite.read_image_hdu(p, hdu=1).array()
    assert int((plain != exp["fits_hcompress_i4_smooth/data"]).sum()) > 500


def test_committed_tile_compressed_files_read_bit_exactly():
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    check_compressed_fits(IO, exp)


def test_survey_tile_loader_on_compressed_bands():
    """``load_fits_bands`` (the host mirror of utils/dataloaders.py:381-447) over tile-compressed band files: the planes are what
    astropy reads, a missing band is a NaN plane, the RA / Dec callback comes from the first band's header."""
    from sky_embeddings_amd.utils.dataloaders import load_fits_bands
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    files = [os.path.join(IO, "astropy_gzip1_f4_dither1.fits"), "None", os.path.join(IO, "astropy_gzip2_f4_lossless.fits")]
    tile, pix_to_radec = load_fits_bands(files, return_wc=True)
    assert tile.shape == (3, 30, 41) and np.isnan(tile[1]).all()
    assert same(tile[0].astype(np.float32), exp["fits_gzip1_f4_dither1/data"].astype(np.float32))
    assert same(tile[2].astype(np.float32), exp["fits_gzip2_f4_lossless/data"].astype(np.float32))
    ra, dec = pix_to_radec(np.array([0.0, 40.0]), np.array([0.0, 29.0]))
    assert np.all(np.isfinite(ra)) and np.all(np.abs(dec - 2.2057) < 0.01)
    tile, _ = load_fits_bands([os.path.join(IO, "astropy_rice_f4_dither1.fits")])
    assert same(tile[0].astype(np.float32), exp["fits_rice_f4_dither1/data"].astype(np.float32))


def test_tile_compressed_nulls_and_refusals(tmp_path):
    """What no astropy 4.3 file exercises (its writer does no null checking): ZBLANK -> NaN, as keyword and as column, on this
    package's own quantising writer; codecs outside the subset and damaged Rice streams are refused loudly."""
    rng = np.random.default_rng(5)
    img = rng.standard_normal((9, 14)).astype(np.float32)
    img[2, 3] = img[7, 0] = np.nan
    for codec in ("GZIP_1", "GZIP_2"):
        p = fits_lite.write_compressed_image_fits(str(tmp_path / f"q_{codec}.fits"), img, codec=codec, tile_rows=4, quantise=1.0 / 64,
                                                  blank_column=codec == "GZIP_2")
        got = fits_lite.read_image_hdu(p, hdu=1).array()
        assert got.dtype == np.float32 and np.array_equal(np.isnan(got), np.isnan(img))
        assert np.nanmax(np.abs(got - img)) <= 0.5 / 64 + 1e-6
    # an astropy Rice file with its codec renamed / its heap cut short
    src = os.path.join(IO, "astropy_rice_i2.fits")
    raw = bytearray(open(src, "rb").read())
    p = str(tmp_path / "h.fits")
    open(p, "wb").write(bytes(raw).replace(b"ZCMPTYPE= 'RICE_1  '", b"ZCMPTYPE= 'BZIP2_1 '"))
    with pytest.raises(NotImplementedError, match="BZIP2_1"):
        fits_lite.read_image_hdu(p, hdu=1)
    hdu = fits_lite.read_image_hdu(src, hdu=1)                          # (sanity: the untouched file reads)
    assert hdu.shape == (37, 53)
    buf = np.frombuffer(bytes(raw), dtype=np.uint8)
    _, pos = fits_lite._read_header(buf, 0)
    hdr, data_pos = fits_lite._read_header(buf, pos)
    heap = data_pos + hdr["NAXIS1"] * hdr["NAXIS2"]
    cut = bytearray(raw)
    cut[heap + 8:heap + hdr["PCOUNT"]] = b"\xff" * (hdr["PCOUNT"] - 8)   # every stream but the start of the first: no code can end
    p = str(tmp_path / "cut.fits")
    open(p, "wb").write(bytes(cut))
    with pytest.raises((RuntimeError, ValueError), match="Rice|rice"):
        fits_lite.read_image_hdu(p, hdu=1)
    for name, what in (("hcompress_i2", "HCOMPRESS"), ("plio_i4", "PLIO")):      # the same damage to the other native codecs
        raw = bytearray(open(os.path.join(IO, f"astropy_{name}.fits"), "rb").read())
        buf = np.frombuffer(bytes(raw), dtype=np.uint8)
        _, pos = fits_lite._read_header(buf, 0)
        hdr, data_pos = fits_lite._read_header(buf, pos)
        heap = data_pos + hdr["NAXIS1"] * hdr["NAXIS2"]
        raw[heap:heap + 64] = b"\x7f" * 64
        p = str(tmp_path / f"cut_{name}.fits")
        open(p, "wb").write(bytes(raw))
        with pytest.raises((RuntimeError, ValueError), match=what):
            fits_lite.read_image_hdu(p, hdu=1)


def test_committed_h5py_files_read_bit_exactly(tmp_path):
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    assert "h5py 3.3.0" in list(exp["versions"])
    check_h5(IO, exp, tmp_path)


def test_committed_h5py_filtered_file_reads_bit_exactly(tmp_path):
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    check_h5_filtered(IO, exp, tmp_path)


def test_committed_astropy_files_and_world_coordinates():
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    assert "astropy 4.3.1" in list(exp["versions"])
    check_fits(IO, exp)


def test_feeder_source_on_an_h5py_chunked_file(tmp_path):
    """The feeder's one-off un-chunking (``Dataset._array`` -> contiguous side file) on a file h5py wrote."""
    exp = np.load(os.path.join(IO, "io_expected.npz"))
    p = str(tmp_path / "r.h5")
    shutil.copy(os.path.join(IO, "h5py_resizable.h5"), p)
    with hdf5_lite.File(p) as f:
        a = f["cutouts"]._array()
        assert same(np.asarray(a), exp["resizable/cutouts"])


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no /opt/conda interpreter with h5py / astropy in this image")
def test_live_files_in_the_reference_geometry(tmp_path):
    """Same checks on files made NOW by the image's own h5py / astropy in the reference's real geometry
    ([n, 5, 64, 64] cutouts -> chunks (128, 1, 8, 16); too large to commit)."""
    out = str(tmp_path / "io")
    env = {"PATH": "/opt/conda/bin:/usr/bin:/bin"}
    r = subprocess.run([CONDA_PY, os.path.join(ROOT, "tests", "golden", "make_io_fixtures.py"), "--out", out, "--full"],
                       env=env, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "ModuleNotFoundError" in r.stderr:
        pytest.skip("conda interpreter lacks h5py / astropy: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr[-2000:]
    exp = np.load(os.path.join(out, "io_expected.npz"))
    assert tuple(int(v) for v in exp["resizable/chunks"]) == (128, 1, 8, 16)
    check_h5(out, exp, tmp_path)
    check_h5_filtered(out, exp, tmp_path)
    check_fits(out, exp)
    check_compressed_fits(out, exp)
