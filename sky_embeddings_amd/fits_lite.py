"""Minimal FITS access for the survey-tile input path (utils/dataloaders.py:330-449: ``fits.open(fn)[1].data`` and
``WCS(hdul[1].header).all_pix2world``) -- astropy is not a dependency of this package.

* ``read_image_hdu(path, hdu=1)``: header (dict) + the RAW big-endian image bytes as a read-only ``numpy.memmap`` and the
  information needed to decode them (BITPIX, BSCALE, BZERO).  The tile sampler copies those bytes to the GPU as they are
  and decodes them there (``skyemb_tile_cutouts``): no host pass over a 70 MB band.  ``image_array`` decodes on the host
  (tests, oracle).
* ``TanSipWCS``: pixel -> (RA, Dec) for the projections the HSC pipeline writes: gnomonic (``RA---TAN`` / ``DEC--TAN``)
  with the CD (or PC + CDELT) matrix and the optional SIP distortion polynomials (``-SIP`` suffix, ``A_p_q`` / ``B_p_q``),
  restated from the FITS WCS papers (Greisen & Calabretta 2002, Calabretta & Greisen 2002) and the SIP convention
  (Shupe et al. 2005).  ``all_pix2world(x, y, origin)`` has astropy's argument meaning.
* ``write_image_fits``: writer for the same subset (tests, synthetic tiles).

Tile-compressed images (``ZIMAGE`` binary tables, what ``fpack`` / astropy ``CompImageHDU`` / the LSST stack write; FITS 4.0
section 10) are decoded on the host into a big-endian float32 / integer array: ``RICE_1`` (the default of both writers),
``PLIO_1`` and ``HCOMPRESS_1`` (native threads over the tiles, ``skyemb_fits_decode_tiles_host``), ``GZIP_1``, ``GZIP_2``
(byte-shuffled) and ``NOCOMPRESS``, for integer
images and for quantised floating-point images (``ZSCALE`` / ``ZZERO`` columns; ``NO_DITHER``, ``SUBTRACTIVE_DITHER_1`` / ``_2`` with
the convention's random sequence; ``ZBLANK`` -> NaN; tiles the writer left unquantised in ``GZIP_COMPRESSED_DATA``).
Pinned against astropy-written files
(``tests/golden/io``).
"""
from __future__ import annotations

import math
import os

import numpy as np

BLOCK = 2880
CARD = 80
_BITPIX_DTYPE = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def _parse_value(raw: str):
    raw = raw.strip()
    if not raw:
        return None
    if raw.startswith("'"):
        end = 1
        out = []
        while end < len(raw):            # '' inside a string is an escaped quote
            if raw[end] == "'":
                if end + 1 < len(raw) and raw[end + 1] == "'":
                    out.append("'")
                    end += 2
                    continue
                break
            out.append(raw[end])
            end += 1
        return "".join(out).rstrip()
    val = raw.split("/", 1)[0].strip()
    if val in ("T", "F"):
        return val == "T"
    try:
        return int(val)
    except ValueError:
        try:
            return float(val.replace("D", "E"))
        except ValueError:
            return val


def _read_header(buf, pos):
    """-> (header dict, position of the first byte after the header blocks)."""
    hdr = {}
    while True:
        block = bytes(buf[pos:pos + BLOCK])
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        pos += BLOCK
        for i in range(0, BLOCK, CARD):
            card = block[i:i + CARD].decode("ascii", errors="replace")
            key = card[:8].strip()
            if key == "END":
                return hdr, pos
            if not key or key in ("COMMENT", "HISTORY") or card[8:10] != "= ":
                continue
            hdr[key] = _parse_value(card[10:])


def _data_bytes(hdr):
    naxis = int(hdr.get("NAXIS", 0))
    if naxis == 0:
        return 0
    n = abs(int(hdr["BITPIX"])) // 8
    for k in range(1, naxis + 1):
        n *= int(hdr[f"NAXIS{k}"])
    n = (n + int(hdr.get("PCOUNT", 0))) * int(hdr.get("GCOUNT", 1))
    return n


class ImageHDU:
    """Header + raw big-endian pixels of one image HDU.  ``raw`` is a read-only memmap [H, W] of the on-disk dtype (or, for a
    decompressed image, an in-memory big-endian array)."""

    def __init__(self, path, header, offset, decoded=None):
        self.path, self.header, self.offset = path, header, offset
        self.bitpix = int(header["BITPIX"])
        if decoded is not None:
            self.shape = decoded.shape
            self.bscale, self.bzero = float(header.get("BSCALE", 1.0)), float(header.get("BZERO", 0.0))
            self.raw = decoded
            return
        if int(header.get("NAXIS", 0)) != 2:
            raise NotImplementedError(f"{path}: image HDU with NAXIS = {header.get('NAXIS')} (2-D images only)")
        self.shape = (int(header["NAXIS2"]), int(header["NAXIS1"]))        # FITS axis 1 is the fastest: [rows, columns]
        self.bscale, self.bzero = float(header.get("BSCALE", 1.0)), float(header.get("BZERO", 0.0))
        self.raw = np.memmap(path, dtype=_BITPIX_DTYPE[self.bitpix], mode="c", offset=offset, shape=self.shape)   # copy-on-write: never written

    def array(self):
        """Decoded pixels, float32 for floating-point images (what astropy's ``.data`` holds, in native byte order)."""
        a = np.asarray(self.raw)
        if self.bitpix < 0:
            out = a.astype(np.float32 if self.bitpix == -32 else np.float64)
        else:
            out = a.astype(np.float64)
        if self.bscale != 1.0 or self.bzero != 0.0:
            out = out * self.bscale + self.bzero
        return out


def read_image_hdu(path, hdu=1) -> ImageHDU:
    """The ``hdu``-th header-data unit of ``path`` (0 = primary), which must be an uncompressed 2-D image."""
    size = os.path.getsize(path)
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    pos, index = 0, 0
    while pos < size:
        hdr, data_pos = _read_header(buf, pos)
        if index == 0 and hdr.get("SIMPLE") is not True and "SIMPLE" not in hdr:
            pass      # astropy is asked to ignore a missing SIMPLE card too (utils/dataloaders.py:417)
        if index == hdu:
            if hdr.get("ZIMAGE") or (hdr.get("XTENSION") == "BINTABLE" and "ZCMPTYPE" in hdr):
                return _read_compressed_image(path, hdr, buf, data_pos)
            if index > 0 and hdr.get("XTENSION") != "IMAGE":
                raise NotImplementedError(f"{path}: HDU {hdu} is a {hdr.get('XTENSION')} extension, not an image")
            return ImageHDU(path, hdr, data_pos)
        nbytes = _data_bytes(hdr)
        pos = data_pos + (nbytes + BLOCK - 1) // BLOCK * BLOCK
        index += 1
    raise IndexError(f"{path}: no HDU {hdu}")


# ---------------------------------------------------------------------------------------------------------------
# tile-compressed images (FITS 4.0 section 10): RICE_1 and the gzip codecs, integer or quantised floating-point pixels
# ---------------------------------------------------------------------------------------------------------------
N_RANDOM = 10000
_ZERO_VALUE = -2147483646          # SUBTRACTIVE_DITHER_2: this quantised value stands for exactly 0.0 (FITS 4.0 section 10.2.3)
_rand_cache = []


def dither_sequence():
    """The 10 000 pseudo-random numbers of the tile-compression convention (FITS 4.0 appendix I): the Park-Miller minimal
    standard generator (a = 16807, m = 2^31 - 1, seed 1) run in double precision, value = seed / m kept in SINGLE precision
    (what the writers' library holds: the dithered pixels of astropy-written files are reproduced bit for bit only with it)."""
    if not _rand_cache:
        a, m, seed = 16807.0, 2147483647.0, 1.0
        out = np.empty(N_RANDOM, dtype=np.float64)
        for i in range(N_RANDOM):
            temp = a * seed
            seed = temp - m * math.floor(temp / m)
            out[i] = seed / m
        _rand_cache.append(out.astype(np.float32))
    return _rand_cache[0]


def _dither_indices(tile_index, zdither0, n):
    """Indices into dither_sequence() for the n pixels of tile `tile_index` (0-based table row): the walk starts at
    int(rand[(row + ZDITHER0 - 1) % 10000] * 500) and, on reaching the end of the table, restarts from the NEXT seed entry."""
    rand = dither_sequence()
    iseed = (tile_index + zdither0 - 1) % N_RANDOM
    nxt = int(rand[iseed] * np.float32(500))
    parts, left = [], n
    while left > 0:
        run = min(left, N_RANDOM - nxt)
        parts.append(np.arange(nxt, nxt + run))
        left -= run
        if left > 0 or nxt + run == N_RANDOM:
            iseed = (iseed + 1) % N_RANDOM
            nxt = int(rand[iseed] * np.float32(500))
    return np.concatenate(parts) if len(parts) > 1 else parts[0]


_CODEC = {"RICE_1": 1, "RICE_ONE": 1, "PLIO_1": 2, "HCOMPRESS_1": 3}


def _decode_tiles(codec, buf, offs, lens, npix, bytepix, blocksize):
    """All Rice / PLIO / HCOMPRESS tiles of an image -> one flat native-endian unsigned array (tile after tile):
    skyemb_fits_decode_tiles_host."""
    import ctypes
    from ._lib import check, lib
    offs, lens, npix = (np.ascontiguousarray(a, dtype=np.int64) for a in (offs, lens, npix))
    dst_off = np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64) if len(npix) else np.zeros(0, np.int64)
    total = int(npix.sum())
    out = np.empty(total, dtype={1: np.uint8, 2: np.uint16, 4: np.uint32}[bytepix])
    base = np.ascontiguousarray(buf)                      # (a memmap stays a view of the mapping)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    check(lib().skyemb_fits_decode_tiles_host(_CODEC[codec], ptr(base), base.size, ptr(offs), ptr(lens), ptr(npix), ptr(dst_off), len(npix),
                                              bytepix, blocksize, ptr(out), total, min(16, os.cpu_count() or 1)),
          "skyemb_fits_decode_tiles_host")
    return out, dst_off


def _dequantise(out, q_parts, tiles, y0, x0, hs, ws, zscale, zzero, zblank_col, zblank_key, method, zdither0):
    """Quantised tiles -> the big-endian float image `out` (skyemb_fits_dequantise_tiles_host, threads over tiles)."""
    import ctypes
    from ._lib import check, lib
    H, W = out.shape
    if isinstance(q_parts, tuple) and out.dtype != np.dtype(">f4"):     # (flat array, offsets) -> per-tile views
        flat, offs = q_parts
        q_parts = [flat[o:o + int(hs[t]) * int(ws[t])] for o, t in zip(offs, tiles)]
    if out.dtype != np.dtype(">f4"):                          # ZBITPIX = -64: rare; the same arithmetic in numpy, tile by tile
        rand = dither_sequence()
        for part, t in zip(q_parts, tiles):
            q = part.astype(np.int64)
            if method.startswith("SUBTRACTIVE"):
                vals = (q.astype(np.float64) - rand[_dither_indices(int(t), zdither0, q.size)].astype(np.float64) + 0.5) * zscale[t] + zzero[t]
                if method.endswith("2"):
                    vals[q == _ZERO_VALUE] = 0.0
            else:
                vals = q.astype(np.float64) * zscale[t] + zzero[t]
            blank = zblank_col[t] if zblank_col is not None else zblank_key
            if blank is not None:
                vals[q == int(blank)] = np.nan
            out[y0[t]:y0[t] + hs[t], x0[t]:x0[t] + ws[t]] = vals.reshape(int(hs[t]), int(ws[t]))
        return
    if isinstance(q_parts, tuple):
        q, q_off = np.ascontiguousarray(q_parts[0], dtype=np.int32), np.ascontiguousarray(q_parts[1], dtype=np.int64)
    else:
        sizes = np.array([p.size for p in q_parts], dtype=np.int64)
        q = np.concatenate([np.asarray(p).astype(np.int32, copy=False) for p in q_parts]) if len(q_parts) > 1 else np.ascontiguousarray(q_parts[0], dtype=np.int32)
        q_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    i64 = lambda a: np.ascontiguousarray(np.asarray(a)[tiles], dtype=np.int64)
    f64 = lambda a: np.ascontiguousarray(np.asarray(a)[tiles], dtype=np.float64)
    a_y0, a_x0, a_h, a_w, rows = i64(y0), i64(x0), i64(hs), i64(ws), np.ascontiguousarray(tiles, dtype=np.int64)
    sc, ze = f64(zscale), f64(zzero)
    if zblank_col is not None:
        blank, has = np.ascontiguousarray(zblank_col[tiles], dtype=np.int32), np.ones(len(tiles), np.uint8)
    elif zblank_key is not None:
        blank, has = np.full(len(tiles), int(zblank_key), np.int32), np.ones(len(tiles), np.uint8)
    else:
        blank = has = None
    code = {"SUBTRACTIVE_DITHER_1": 1, "SUBTRACTIVE_DITHER_2": 2}.get(method, 0)
    rand = dither_sequence()
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
    assert out.flags.c_contiguous
    check(lib().skyemb_fits_dequantise_tiles_host(ptr(q), ptr(q_off), ptr(a_y0), ptr(a_x0), ptr(a_h), ptr(a_w), ptr(rows), len(tiles), ptr(sc),
                                                  ptr(ze), ptr(blank), ptr(has), ptr(rand), code, zdither0, ptr(out), H, W, 1,
                                                  min(16, os.cpu_count() or 1)), "skyemb_fits_dequantise_tiles_host")


def _read_compressed_image(path, hdr, buf, data_pos):
    import re
    import zlib
    codec = str(hdr.get("ZCMPTYPE", "")).strip()
    if codec not in ("RICE_1", "RICE_ONE", "GZIP_1", "GZIP_2", "NOCOMPRESS", "PLIO_1", "HCOMPRESS_1"):
        raise NotImplementedError(f"{path}: tile compression {codec!r} is not one of RICE_1, GZIP_1, GZIP_2, PLIO_1, HCOMPRESS_1, NOCOMPRESS")
    rice = codec in _CODEC                                    # (decoded natively, all tiles in one call)
    zbitpix, znaxis = int(hdr["ZBITPIX"]), int(hdr.get("ZNAXIS", 0))
    if znaxis != 2:
        raise NotImplementedError(f"{path}: compressed image with ZNAXIS = {znaxis} (2-D images only)")
    W, H = int(hdr["ZNAXIS1"]), int(hdr["ZNAXIS2"])
    tw, th = int(hdr.get("ZTILE1", W)), int(hdr.get("ZTILE2", 1))
    nfields, row_bytes, nrows = int(hdr["TFIELDS"]), int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    cols, off = {}, 0
    widths = {"L": 1, "B": 1, "I": 2, "J": 4, "K": 8, "E": 4, "D": 8, "A": 1}
    for k in range(1, nfields + 1):
        m = re.match(r"\s*(\d*)([PQ])?([A-Z])", str(hdr[f"TFORM{k}"]))
        if not m:
            raise ValueError(f"{path}: TFORM{k} = {hdr[f'TFORM{k}']!r}")
        repeat = int(m.group(1)) if m.group(1) else 1
        name = str(hdr.get(f"TTYPE{k}", "")).strip()
        if m.group(2):                                   # variable-length array: (length, heap offset) descriptor
            cols[name] = (off, m.group(2), m.group(3))
            off += repeat * (16 if m.group(2) == "Q" else 8)
        else:
            cols[name] = (off, m.group(3), m.group(3))
            off += repeat * widths[m.group(3)]
    if "COMPRESSED_DATA" not in cols and "GZIP_COMPRESSED_DATA" not in cols:
        raise NotImplementedError(f"{path}: compressed image table without a COMPRESSED_DATA column")
    tiles_x, tiles_y = (W + tw - 1) // tw, (H + th - 1) // th
    if nrows != tiles_x * tiles_y:
        raise ValueError(f"{path}: {nrows} table rows for {tiles_x} x {tiles_y} tiles")
    # algorithm parameters (ZNAMEi / ZVALi) and the quantisation of floating-point pixels
    params = {str(hdr[f"ZNAME{i}"]).strip().upper(): hdr.get(f"ZVAL{i}") for i in range(1, 10) if f"ZNAME{i}" in hdr}
    blocksize, bytepix = int(params.get("BLOCKSIZE", 32)), int(params.get("BYTEPIX", 4))
    if codec in ("PLIO_1", "HCOMPRESS_1"):
        bytepix = 4                                           # (both decode to 32-bit integers)
        blocksize = 0
        if codec == "HCOMPRESS_1":
            blocksize = 1 if int(params.get("SMOOTH", 0) or 0) != 0 else 0      # (the decoder's per-codec parameter: smoothing on / off)
    quantised = zbitpix < 0 and "ZSCALE" in cols
    method = str(hdr.get("ZQUANTIZ", "NO_DITHER" if quantised else "NONE")).strip().upper()
    if zbitpix < 0 and rice and not quantised:
        raise ValueError(f"{path}: {codec} floating-point image without ZSCALE / ZZERO columns")
    if method not in ("NONE", "", "NO_DITHER", "SUBTRACTIVE_DITHER_1", "SUBTRACTIVE_DITHER_2"):
        raise NotImplementedError(f"{path}: ZQUANTIZ = {method!r}")
    zdither0 = int(hdr.get("ZDITHER0", 1))
    heap = data_pos + int(hdr.get("THEAP", row_bytes * nrows))
    table = np.asarray(buf[data_pos:data_pos + row_bytes * nrows]).reshape(nrows, row_bytes)

    def column(name):
        o, kind, elem = cols[name]
        if kind in "PQ":                                  # [nrows, 2] = (element count, heap offset)
            n = 16 if kind == "Q" else 8
            return np.ascontiguousarray(table[:, o:o + n]).view(">i8" if kind == "Q" else ">i4").astype(np.int64).reshape(nrows, 2)
        w = widths[kind]
        return np.ascontiguousarray(table[:, o:o + w]).view({"E": ">f4", "D": ">f8", "J": ">i4", "K": ">i8", "I": ">i2", "B": "u1"}[kind]).reshape(nrows)

    desc = {c: column(c) for c in ("COMPRESSED_DATA", "GZIP_COMPRESSED_DATA", "UNCOMPRESSED_DATA") if c in cols}
    zscale = column("ZSCALE").astype(np.float64) if "ZSCALE" in cols else None
    zzero = column("ZZERO").astype(np.float64) if "ZZERO" in cols else None
    if quantised and zzero is None:
        raise ValueError(f"{path}: ZSCALE column without ZZERO")
    zblank_col = column("ZBLANK") if "ZBLANK" in cols else None
    zblank_key = hdr.get("ZBLANK")
    # what the main column's bytes decode to: quantised floats and integer images are integers of ZBITPIX (quantised: 32) bits
    q_dt = np.dtype(">i4") if quantised else np.dtype(_BITPIX_DTYPE[zbitpix])
    out_dt = np.dtype(_BITPIX_DTYPE[zbitpix])
    out = np.empty((H, W), dtype=out_dt)
    ty, tx = np.divmod(np.arange(nrows), tiles_x)
    hs, ws = np.minimum(th, H - ty * th), np.minimum(tw, W - tx * tw)
    npix = (hs * ws).astype(np.int64)
    main = desc.get("COMPRESSED_DATA")
    in_main = main[:, 0] > 0 if main is not None else np.zeros(nrows, bool)
    rice_out = rice_off = None
    if rice and in_main.any():
        if bytepix not in (1, 2, 4):
            raise NotImplementedError(f"{path}: RICE_1 with BYTEPIX = {bytepix}")
        sel = np.nonzero(in_main)[0]
        words = 2 if cols["COMPRESSED_DATA"][2] == "I" else 1      # PLIO_1: 16-bit words (1PI); the others are byte arrays
        rice_out, offs_sel = _decode_tiles(codec, buf, heap + main[sel, 1], main[sel, 0] * words, npix[sel], bytepix, blocksize)
        rice_off = dict(zip(sel.tolist(), offs_sel.tolist()))
    signed = {1: np.uint8, 2: np.int16, 4: np.int32}          # (8-bit FITS pixels are unsigned)
    q_parts, q_tiles = [], []                                 # quantised tiles: dequantised together below
    for t in range(nrows):
        h, w, n = int(hs[t]), int(ws[t]), int(npix[t])
        dest = out[ty[t] * th:ty[t] * th + h, tx[t] * tw:tx[t] * tw + w]
        if in_main[t]:
            if rice and quantised and bytepix == 4:
                continue                                      # (stay where the decoder put them: dequantised in one call below)
            if rice:
                ints = rice_out[rice_off[t]:rice_off[t] + n].view(signed[bytepix])
            else:
                length, ptr = int(main[t, 0]), int(main[t, 1])
                chunk = bytes(buf[heap + ptr:heap + ptr + length])
                raw = chunk if codec == "NOCOMPRESS" else zlib.decompress(chunk, 15 + 32)      # zlib or gzip wrapper
                if codec == "GZIP_2":                         # byte planes, most significant first -> pixels
                    raw = np.frombuffer(raw, dtype=np.uint8).reshape(q_dt.itemsize, n).T.tobytes()
                if len(raw) != n * q_dt.itemsize:
                    raise ValueError(f"{path}: tile {t} decompressed to {len(raw)} bytes, expected {n * q_dt.itemsize}")
                ints = np.frombuffer(raw, dtype=q_dt)
            if quantised:
                q_parts.append(ints)
                q_tiles.append(t)
            else:
                dest[...] = ints.reshape(h, w)                # lossless: integers, or floats of the gzip codecs
            continue
        # tiles the writer could not quantise (or chose not to compress) stand as gzip-compressed / raw pixels of ZBITPIX
        for col in ("GZIP_COMPRESSED_DATA", "UNCOMPRESSED_DATA"):
            if col in desc and desc[col][t, 0] > 0:
                length, ptr = int(desc[col][t, 0]), int(desc[col][t, 1])
                if col == "UNCOMPRESSED_DATA":
                    raw = bytes(buf[heap + ptr:heap + ptr + length * out_dt.itemsize])
                else:
                    raw = zlib.decompress(bytes(buf[heap + ptr:heap + ptr + length]), 15 + 32)
                if len(raw) != n * out_dt.itemsize:
                    raise ValueError(f"{path}: tile {t} ({col}) holds {len(raw)} bytes, expected {n * out_dt.itemsize}")
                dest[...] = np.frombuffer(raw, dtype=out_dt).reshape(h, w)
                break
        else:
            raise ValueError(f"{path}: tile {t} has no data in any column")
    if rice and quantised and bytepix == 4 and in_main.any():
        sel = np.nonzero(in_main)[0]
        flat = (rice_out.view(np.int32), np.array([rice_off[int(t)] for t in sel], dtype=np.int64))
        _dequantise(out, flat, sel, ty * th, tx * tw, hs, ws, zscale, zzero, zblank_col, zblank_key, method, zdither0)
    elif q_tiles:
        _dequantise(out, q_parts, np.asarray(q_tiles), ty * th, tx * tw, hs, ws, zscale, zzero, zblank_col, zblank_key, method, zdither0)
    image_hdr = dict(hdr)
    image_hdr["BITPIX"] = zbitpix
    for key in ("BSCALE", "BZERO"):
        if "Z" + key in hdr:
            image_hdr[key] = hdr["Z" + key]
    return ImageHDU(path, image_hdr, None, decoded=out)


def write_compressed_image_fits(path, image, header=None, codec="GZIP_2", tile_rows=1, quantise=None, blank_column=False):
    """Primary HDU + ONE tile-compressed image extension (BINTABLE, COMPRESSED_DATA as 1PB variable-length arrays, row tiles)
    with a gzip codec -- the layout read back by ``read_image_hdu`` (tests).  Lossless by default; ``quantise=scale`` stores a
    float32 image as int32 round((x - zero) / scale) with per-tile ZSCALE / ZZERO columns (ZQUANTIZ = 'NO_DITHER'), NaN as
    ZBLANK (a header keyword, or a column with ``blank_column``)."""
    import zlib
    image = np.asarray(image)
    assert image.ndim == 2 and codec in ("GZIP_1", "GZIP_2")
    bitpix = {np.dtype("float32"): -32, np.dtype("float64"): -64, np.dtype("int16"): 16, np.dtype("int32"): 32}[image.dtype]
    be = image.astype(_BITPIX_DTYPE[bitpix])
    H, W = image.shape
    NULL = -2147483647
    chunks, scales, zeros = [], [], []
    for y in range(0, H, tile_rows):
        if quantise is not None:
            assert bitpix == -32
            tile = image[y:y + tile_rows].astype(np.float64)
            zero = float(np.nanmin(tile)) if np.isfinite(tile).any() else 0.0
            q = np.where(np.isnan(tile), NULL, np.rint((np.nan_to_num(tile) - zero) / quantise)).astype(">i4")
            raw, item = q.tobytes(), 4
            scales.append(float(quantise))
            zeros.append(zero)
        else:
            raw, item = be[y:y + tile_rows].tobytes(), be.itemsize
        if codec == "GZIP_2":
            n = len(raw) // item
            raw = np.frombuffer(raw, dtype=np.uint8).reshape(n, item).T.tobytes()
        chunks.append(zlib.compress(raw, 6))
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]])
    rows = []
    for i, (c, o) in enumerate(zip(chunks, offs)):
        row = np.array([len(c), o], dtype=">i4").tobytes()
        if quantise is not None:
            row += np.array([scales[i], zeros[i]], dtype=">f8").tobytes()
            if blank_column:
                row += np.array([NULL], dtype=">i4").tobytes()
        rows.append(row)
    table = b"".join(rows)
    heap = b"".join(chunks)
    primary = [_card("SIMPLE", True), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True)]
    fields = [("COMPRESSED_DATA", f"1PB({max(len(c) for c in chunks)})")]
    if quantise is not None:
        fields += [("ZSCALE", "1D"), ("ZZERO", "1D")] + ([("ZBLANK", "1J")] if blank_column else [])
    ext = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", len(rows[0])), _card("NAXIS2", len(chunks)),
           _card("PCOUNT", len(heap)), _card("GCOUNT", 1), _card("TFIELDS", len(fields))]
    for i, (name, form) in enumerate(fields, 1):
        ext += [_card(f"TTYPE{i}", name), _card(f"TFORM{i}", form)]
    ext += [_card("ZIMAGE", True), _card("ZCMPTYPE", codec), _card("ZBITPIX", bitpix), _card("ZNAXIS", 2), _card("ZNAXIS1", W),
            _card("ZNAXIS2", H), _card("ZTILE1", W), _card("ZTILE2", tile_rows), _card("ZQUANTIZ", "NONE" if quantise is None else "NO_DITHER")]
    if quantise is not None and not blank_column:
        ext.append(_card("ZBLANK", NULL))
    for k, v in (header or {}).items():
        ext.append(_card(k, v))
    data = table + heap
    data += b"\0" * (-len(data) % BLOCK)
    with open(path, "wb") as f:
        f.write(_header_bytes(primary))
        f.write(_header_bytes(ext))
        f.write(data)
    return path


# ---------------------------------------------------------------------------------------------------------------
# writer (tests / synthetic tiles)
# ---------------------------------------------------------------------------------------------------------------
def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, int):
        v = f"{value:>20d}"
    elif isinstance(value, float):
        v = f"{value:>20.13E}" if value != 0 else f"{'0.0':>20}"
    else:
        v = "'" + f"{str(value):<8}".replace("'", "''") + "'"
        v = f"{v:<20}"
    card = f"{key:<8}= {v}"
    if comment:
        card += f" / {comment}"
    return f"{card:<80}"[:80]


def _header_bytes(cards):
    text = "".join(cards) + f"{'END':<80}"
    text += " " * (-len(text) % BLOCK)
    return text.encode("ascii")


def write_image_fits(path, image, header=None, bitpix=-32):
    """Primary HDU without data + ONE image extension (the layout ``hdul[1].data`` expects).  ``header``: extra cards of
    the image extension (e.g. the WCS keywords), values bool / int / float / str."""
    image = np.asarray(image)
    assert image.ndim == 2
    primary = [_card("SIMPLE", True), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True)]
    ext = [_card("XTENSION", "IMAGE"), _card("BITPIX", int(bitpix)), _card("NAXIS", 2), _card("NAXIS1", int(image.shape[1])),
           _card("NAXIS2", int(image.shape[0])), _card("PCOUNT", 0), _card("GCOUNT", 1)]
    for k, v in (header or {}).items():
        ext.append(_card(k, v))
    data = np.ascontiguousarray(image.astype(_BITPIX_DTYPE[bitpix])).tobytes()
    data += b"\0" * (-len(data) % BLOCK)
    with open(path, "wb") as f:
        f.write(_header_bytes(primary))
        f.write(_header_bytes(ext))
        f.write(data)
    return path


# ---------------------------------------------------------------------------------------------------------------
# world coordinates: TAN (+ SIP)
# ---------------------------------------------------------------------------------------------------------------
class TanSipWCS:
    """Pixel -> celestial coordinates for ``CTYPE = RA---TAN[-SIP], DEC--TAN[-SIP]`` headers.

    pixel (FITS 1-based p1, p2) -> u = p - CRPIX -> SIP: u' = u + A(u, v), v' = v + B(u, v) -> intermediate world
    coordinates (x, y) = CD . (u', v') [deg] -> native spherical (phi, theta) of the gnomonic projection ->
    celestial (RA, Dec) with the pole at (CRVAL1, CRVAL2), LONPOLE = 180 deg (the zenithal default for theta0 = 90)."""

    def __init__(self, header):
        c1, c2 = str(header.get("CTYPE1", "")), str(header.get("CTYPE2", ""))
        if not (c1.startswith("RA---TAN") and c2.startswith("DEC--TAN")):
            raise NotImplementedError(f"WCS projection {c1!r}, {c2!r}: only RA---TAN / DEC--TAN (with or without -SIP)")
        self.crpix = (float(header["CRPIX1"]), float(header["CRPIX2"]))
        self.crval = (float(header["CRVAL1"]), float(header["CRVAL2"]))
        if "CD1_1" in header:
            self.cd = np.array([[float(header.get("CD1_1", 0.0)), float(header.get("CD1_2", 0.0))],
                                [float(header.get("CD2_1", 0.0)), float(header.get("CD2_2", 0.0))]])
        else:
            pc = np.array([[float(header.get("PC1_1", 1.0)), float(header.get("PC1_2", 0.0))],
                           [float(header.get("PC2_1", 0.0)), float(header.get("PC2_2", 1.0))]])
            self.cd = np.diag([float(header.get("CDELT1", 1.0)), float(header.get("CDELT2", 1.0))]) @ pc
        self.sip = c1.endswith("-SIP")
        self.a, self.b = {}, {}
        if self.sip:
            for name, dst in (("A", self.a), ("B", self.b)):
                order = int(header.get(f"{name}_ORDER", 0))
                for p in range(order + 1):
                    for q in range(order + 1 - p):
                        v = header.get(f"{name}_{p}_{q}")
                        if v is not None and float(v) != 0.0:
                            dst[(p, q)] = float(v)

    def all_pix2world(self, x, y, origin):
        """astropy's ``WCS.all_pix2world(x, y, origin)``: x = FITS axis 1 (columns), y = axis 2 (rows), 0- or 1-based.
        -> (ra, dec) in degrees, float64 arrays."""
        x = np.asarray(x, dtype=np.float64) + (1 - origin)
        y = np.asarray(y, dtype=np.float64) + (1 - origin)
        u, v = x - self.crpix[0], y - self.crpix[1]
        if self.sip:
            du = sum(c * u ** p * v ** q for (p, q), c in self.a.items()) if self.a else 0.0
            dv = sum(c * u ** p * v ** q for (p, q), c in self.b.items()) if self.b else 0.0
            u, v = u + du, v + dv
        xi = np.deg2rad(self.cd[0, 0] * u + self.cd[0, 1] * v)
        eta = np.deg2rad(self.cd[1, 0] * u + self.cd[1, 1] * v)
        # gnomonic de-projection about the reference point (alpha0, delta0): the standard-coordinate form of
        # Calabretta & Greisen (2002) eqs. 14-15, 54-55 with phi_p = 180 deg
        a0, d0 = math.radians(self.crval[0]), math.radians(self.crval[1])
        den = math.cos(d0) - eta * math.sin(d0)
        ra = a0 + np.arctan2(xi, den)
        dec = np.arctan2(math.sin(d0) + eta * math.cos(d0), np.hypot(xi, den))
        ra = np.mod(np.rad2deg(ra), 360.0)
        return ra, np.rad2deg(dec)
