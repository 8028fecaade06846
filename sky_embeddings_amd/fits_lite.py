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
section 10): the lossless gzip codecs (``GZIP_1``, ``GZIP_2`` = byte-shuffled, ``NOCOMPRESS``) without quantisation are
decoded on the host (zlib) into a native float32 / integer array; ``RICE_1`` / ``HCOMPRESS_1`` / ``PLIO_1`` and quantised
floating-point tiles (``ZSCALE`` / ``ZZERO`` columns, dithering) raise ``NotImplementedError`` by name: funpack such files.
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
# tile-compressed images (FITS 4.0 section 10), lossless gzip codecs
# ---------------------------------------------------------------------------------------------------------------
def _read_compressed_image(path, hdr, buf, data_pos):
    import zlib
    codec = str(hdr.get("ZCMPTYPE", "")).strip()
    if codec not in ("GZIP_1", "GZIP_2", "NOCOMPRESS"):
        raise NotImplementedError(f"{path}: tile compression {codec!r} is not supported (lossless GZIP_1 / GZIP_2 only): funpack the file")
    zbitpix, znaxis = int(hdr["ZBITPIX"]), int(hdr.get("ZNAXIS", 0))
    if znaxis != 2:
        raise NotImplementedError(f"{path}: compressed image with ZNAXIS = {znaxis} (2-D images only)")
    W, H = int(hdr["ZNAXIS1"]), int(hdr["ZNAXIS2"])
    tw, th = int(hdr.get("ZTILE1", W)), int(hdr.get("ZTILE2", 1))
    nfields, row_bytes, nrows = int(hdr["TFIELDS"]), int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    cols, off = {}, 0
    import re
    widths = {"L": 1, "B": 1, "I": 2, "J": 4, "K": 8, "E": 4, "D": 8, "A": 1}
    for k in range(1, nfields + 1):
        m = re.match(r"\s*(\d*)([PQ])?([A-Z])", str(hdr[f"TFORM{k}"]))
        if not m:
            raise ValueError(f"{path}: TFORM{k} = {hdr[f'TFORM{k}']!r}")
        repeat = int(m.group(1)) if m.group(1) else 1
        name = str(hdr.get(f"TTYPE{k}", "")).strip()
        if m.group(2):                                   # variable-length array: (length, heap offset) descriptor
            cols[name] = (off, m.group(2))
            off += repeat * (16 if m.group(2) == "Q" else 8)
        else:
            cols[name] = (off, m.group(3))
            off += repeat * widths[m.group(3)]
    if zbitpix < 0 and ("ZSCALE" in cols or str(hdr.get("ZQUANTIZ", "NONE")).strip().upper() not in ("NONE", "")):
        raise NotImplementedError(f"{path}: quantised floating-point tiles ({hdr.get('ZQUANTIZ')}) are not supported: funpack the file")
    if "COMPRESSED_DATA" not in cols and "GZIP_COMPRESSED_DATA" not in cols:
        raise NotImplementedError(f"{path}: compressed image table without a COMPRESSED_DATA column")
    heap = data_pos + int(hdr.get("THEAP", row_bytes * nrows))
    dt = np.dtype(_BITPIX_DTYPE[zbitpix])
    out = np.empty((H, W), dtype=dt)
    tiles_x = (W + tw - 1) // tw
    table = np.asarray(buf[data_pos:data_pos + row_bytes * nrows]).reshape(nrows, row_bytes)

    def descriptor(row, col):
        o, kind = cols[col]
        n = 16 if kind == "Q" else 8
        vals = np.frombuffer(table[row, o:o + n].tobytes(), dtype=">i8" if kind == "Q" else ">i4")
        return int(vals[0]), int(vals[1])

    for t in range(nrows):
        ty, tx = divmod(t, tiles_x)
        h, w = min(th, H - ty * th), min(tw, W - tx * tw)
        nbytes = h * w * dt.itemsize
        length, ptr, col = 0, 0, None
        for col in ("COMPRESSED_DATA", "GZIP_COMPRESSED_DATA", "UNCOMPRESSED_DATA"):
            if col in cols:
                length, ptr = descriptor(t, col)
                if length > 0:
                    break
        chunk = bytes(buf[heap + ptr:heap + ptr + length * (dt.itemsize if col == "UNCOMPRESSED_DATA" else 1)])
        if col == "UNCOMPRESSED_DATA" or codec == "NOCOMPRESS":
            raw = chunk
        else:
            raw = zlib.decompress(chunk, 15 + 32)              # zlib or gzip wrapper
            if codec == "GZIP_2" and col == "COMPRESSED_DATA":     # byte planes, most significant first -> pixels
                raw = np.frombuffer(raw, dtype=np.uint8).reshape(dt.itemsize, h * w).T.tobytes()
        if len(raw) != nbytes:
            raise ValueError(f"{path}: tile {t} decompressed to {len(raw)} bytes, expected {nbytes}")
        out[ty * th:ty * th + h, tx * tw:tx * tw + w] = np.frombuffer(raw, dtype=dt).reshape(h, w)
    image_hdr = dict(hdr)
    image_hdr["BITPIX"] = zbitpix
    for key in ("BSCALE", "BZERO"):
        if "Z" + key in hdr:
            image_hdr[key] = hdr["Z" + key]
    return ImageHDU(path, image_hdr, None, decoded=out)


def write_compressed_image_fits(path, image, header=None, codec="GZIP_2", tile_rows=1):
    """Primary HDU + ONE tile-compressed image extension (BINTABLE, COMPRESSED_DATA as 1PB variable-length arrays, row tiles)
    with a lossless gzip codec -- the layout read back by ``read_image_hdu`` (tests)."""
    import zlib
    image = np.asarray(image)
    assert image.ndim == 2 and codec in ("GZIP_1", "GZIP_2")
    bitpix = {np.dtype("float32"): -32, np.dtype("float64"): -64, np.dtype("int16"): 16, np.dtype("int32"): 32}[image.dtype]
    be = image.astype(_BITPIX_DTYPE[bitpix])
    H, W = image.shape
    chunks = []
    for y in range(0, H, tile_rows):
        raw = be[y:y + tile_rows].tobytes()
        if codec == "GZIP_2":
            n = len(raw) // be.itemsize
            raw = np.frombuffer(raw, dtype=np.uint8).reshape(n, be.itemsize).T.tobytes()
        chunks.append(zlib.compress(raw, 6))
    offs = np.cumsum([0] + [len(c) for c in chunks[:-1]])
    table = b"".join(np.array([len(c), o], dtype=">i4").tobytes() for c, o in zip(chunks, offs))
    heap = b"".join(chunks)
    primary = [_card("SIMPLE", True), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True)]
    ext = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", 8), _card("NAXIS2", len(chunks)),
           _card("PCOUNT", len(heap)), _card("GCOUNT", 1), _card("TFIELDS", 1), _card("TTYPE1", "COMPRESSED_DATA"),
           _card("TFORM1", f"1PB({max(len(c) for c in chunks)})"), _card("ZIMAGE", True), _card("ZCMPTYPE", codec),
           _card("ZBITPIX", bitpix), _card("ZNAXIS", 2), _card("ZNAXIS1", W), _card("ZNAXIS2", H), _card("ZTILE1", W),
           _card("ZTILE2", tile_rows), _card("ZQUANTIZ", "NONE")]
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
