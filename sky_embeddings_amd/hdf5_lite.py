"""Minimal HDF5 access for the cutout files of the hot path (no h5py on the build / GPU boxes).

Schema (reference ``data_processing/utils.py:333-361``, ``configs/README.md:10-13``): root-level
datasets ``cutouts float32 [N,C,H,W]``, ``ra``/``dec`` float32 [N], optional ``zspec``,
``zspec_err``, ``class`` -- created with ``create_dataset(name, shape, dtype)`` i.e. CONTIGUOUS
layout, no filters.  That subset of the HDF5 1.x file format is what this module reads
(superblock v0/v1, v1 object headers incl. continuation blocks, v1 group B-trees + local heaps,
dataspace v1/v2, fixed-point / IEEE float little-endian datatypes, contiguous (and compact) data
layout v3) and writes (the synthetic-data generator).  Chunked / compressed / new-style-group
files raise ``NotImplementedError`` naming the feature.  Datasets are exposed as ``numpy.memmap``
views, so a [N,5,64,64] cutout file is read with plain coalesced page-cache I/O instead of the
reference's per-item ``h5py.File`` open (utils/dataloaders.py:289).

If ``h5py`` is importable it is NOT used: one code path everywhere keeps results reproducible.
"""
from __future__ import annotations

import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5LiteError(IOError):
    pass


# ------------------------------------------------------------------------------------------ reader
class Dataset:
    def __init__(self, name, shape, dtype, offset, fileobj_path, inline=None):
        self.name, self.shape, self.dtype = name, tuple(shape), np.dtype(dtype)
        self._offset, self._path, self._inline = offset, fileobj_path, inline
        self._mm = None

    def _array(self):
        if self._mm is None:
            if self._inline is not None:
                self._mm = np.frombuffer(self._inline, dtype=self.dtype).reshape(self.shape)
            elif self._offset == UNDEF or int(np.prod(self.shape)) == 0:
                self._mm = np.zeros(self.shape, self.dtype)  # never written: fill value 0
            else:
                self._mm = np.memmap(self._path, dtype=self.dtype, mode="r", offset=self._offset, shape=self.shape)
        return self._mm

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return np.array(self._array()[idx])  # copy, like h5py (callers mutate: utils/dataloaders.py:294)

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._array())
        return a.astype(dtype) if dtype is not None else a


class File:
    """Read-only view: ``with File(path) as f: f['cutouts'][i]``, ``'ra' in f``, ``f.keys()``."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("hdf5_lite.File is read-only; use write_datasets() to create files")
        self.path = path
        with open(path, "rb") as fh:
            self._buf = fh.read(1 << 20)  # metadata of these files lives in the first MiB
            self._fh_size = fh.seek(0, 2)
        self._datasets = {}
        self._parse()

    # -- low level
    def _read(self, off, n):
        if off + n > len(self._buf):
            with open(self.path, "rb") as fh:
                fh.seek(off)
                return fh.read(n)
        return self._buf[off:off + n]

    def _u(self, off, n):
        return int.from_bytes(self._read(off, n), "little")

    def _parse(self):
        if self._read(0, 8) != SIG:
            raise H5LiteError(f"{self.path}: not an HDF5 file (bad signature)")
        ver = self._u(8, 1)
        if ver not in (0, 1):
            raise NotImplementedError(f"{self.path}: superblock version {ver} (libver='latest' files) is not supported "
                                      "by hdf5_lite; re-save with the default libver")
        so, sl = self._u(13, 1), self._u(14, 1)
        if (so, sl) != (8, 8):
            raise NotImplementedError("only 8-byte offsets/lengths are supported")
        p = 24 if ver == 0 else 28
        self._base = self._u(p, 8)
        root_entry = p + 32
        ohdr = self._u(root_entry + 8, 8)
        cache_type = self._u(root_entry + 16, 4)
        if cache_type == 1:
            btree, heap = self._u(root_entry + 24, 8), self._u(root_entry + 32, 8)
        else:
            btree = heap = None
            for mtype, data in self._messages(ohdr):
                if mtype == 0x0011:
                    btree, heap = struct.unpack("<QQ", data[:16])
            if btree is None:
                raise NotImplementedError("root group without a symbol table (new-style groups) is not supported")
        heap_data = self._heap_data(heap)
        for name_off, obj in self._walk_btree(btree):
            end = heap_data.index(b"\0", name_off)
            name = heap_data[name_off:end].decode()
            ds = self._dataset(name, obj)
            if ds is not None:
                self._datasets[name] = ds

    def _heap_data(self, addr):
        if self._read(addr, 4) != b"HEAP":
            raise H5LiteError("bad local heap signature")
        size, _free, daddr = struct.unpack("<QQQ", self._read(addr + 8, 24))
        return self._read(daddr, size)

    def _walk_btree(self, addr):
        if self._read(addr, 4) != b"TREE":
            raise H5LiteError("bad B-tree signature")
        ntype, level, used = self._u(addr + 4, 1), self._u(addr + 5, 1), self._u(addr + 6, 2)
        if ntype != 0:
            raise H5LiteError("unexpected B-tree node type for a group")
        p = addr + 24
        for i in range(used):
            child = self._u(p + 8 + i * 16, 8)
            if level > 0:
                yield from self._walk_btree(child)
            else:
                if self._read(child, 4) != b"SNOD":
                    raise H5LiteError("bad symbol node signature")
                n = self._u(child + 6, 2)
                for j in range(n):
                    e = child + 8 + j * 40
                    yield self._u(e, 8), self._u(e + 8, 8)

    def _messages(self, addr):
        ver = self._u(addr, 1)
        if ver != 1:
            raise NotImplementedError("version-2 object headers (libver='latest') are not supported by hdf5_lite")
        nmsg = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, remaining = blocks.pop(0)
            end = p + remaining
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = struct.unpack("<HH", self._read(p, 4))
                data = self._read(p + 8, msize)
                if mtype == 0x0010:  # continuation
                    off, length = struct.unpack("<QQ", data[:16])
                    blocks.append((off, length))
                out.append((mtype, data))
                p += 8 + msize
        return out

    def _dataset(self, name, addr):
        shape = dtype = None
        layout = None
        for mtype, d in self._messages(addr):
            if mtype == 0x0001:
                v, rank, flags = d[0], d[1], d[2]
                p = 8 if v == 1 else 4
                shape = struct.unpack("<" + "Q" * rank, d[p:p + 8 * rank])
            elif mtype == 0x0003:
                cls, size = d[0] & 0x0F, struct.unpack("<I", d[4:8])[0]
                big = d[1] & 1
                if big:
                    raise NotImplementedError("big-endian datasets are not supported")
                if cls == 1:
                    dtype = {2: "<f2", 4: "<f4", 8: "<f8"}[size]
                elif cls == 0:
                    dtype = ("<i" if d[1] & 0x08 else "<u") + str(size)
                else:
                    dtype = None  # strings etc.: ignored (not part of the cutout schema)
            elif mtype == 0x0008:
                v = d[0]
                if v != 3:
                    raise NotImplementedError(f"data layout message version {v} is not supported")
                cls = d[1]
                if cls == 1:
                    a, s = struct.unpack("<QQ", d[2:18])
                    layout = ("contiguous", a, s)
                elif cls == 0:
                    n = struct.unpack("<H", d[2:4])[0]
                    layout = ("compact", bytes(d[4:4 + n]), n)
                else:
                    raise NotImplementedError(f"dataset {name!r} is chunked/filtered: hdf5_lite reads contiguous datasets "
                                              "only (the reference writes them with create_dataset(name, shape, dtype))")
            elif mtype == 0x000B:
                raise NotImplementedError(f"dataset {name!r} uses a filter pipeline (compression)")
        if shape is None or layout is None or dtype is None:
            return None  # a sub-group or an unsupported type: not part of the schema
        if layout[0] == "compact":
            return Dataset(name, shape, dtype, None, self.path, inline=layout[1])
        a = layout[1]
        return Dataset(name, shape, dtype, a if a == UNDEF else a + self._base, self.path)

    # -- mapping interface
    def __getitem__(self, k):
        return self._datasets[k]

    def __contains__(self, k):
        return k in self._datasets

    def keys(self):
        return list(self._datasets)

    def close(self):
        self._datasets.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------------------------------ writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            sign = 31
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            sign = 63
        head = bytes([0x11, 0x20, sign, 0x00]) + struct.pack("<I", dt.itemsize)
        return head + props
    if dt.kind in "iu":
        head = bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0x00, 0x00]) + struct.pack("<I", dt.itemsize)
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    raise TypeError(f"hdf5_lite cannot write dtype {dt}")


def write_datasets(path, datasets: dict):
    """Create an HDF5 file with root-level contiguous datasets (h5py-readable, 'earliest' format).
    ``datasets`` maps name -> ndarray (written little-endian, C order)."""
    names = sorted(datasets)
    if len(names) > 8:
        raise ValueError("hdf5_lite writes at most 8 root datasets (one symbol-table node)")
    arrays = {n: np.ascontiguousarray(datasets[n]) for n in names}
    # ---- layout plan ------------------------------------------------------------------------
    sb_size = 24 + 32 + 40           # superblock v0 + root symbol table entry
    root_ohdr = sb_size               # 96
    root_msgs = _msg(0x0011, struct.pack("<QQ", 0, 0))  # patched below
    root_ohdr_size = 16 + len(root_msgs)
    btree = root_ohdr + root_ohdr_size
    K_int, K_leaf = 16, 4
    btree_size = 24 + (2 * K_int + 1) * 8 + 2 * K_int * 8
    heap = btree + btree_size
    heap_data = b"\0" * 8
    name_off = {}
    for n in names:
        name_off[n] = len(heap_data)
        heap_data += _pad8(n.encode() + b"\0")
    heap_data += b"\0" * (-len(heap_data) % 8)
    if len(heap_data) < 88:
        heap_data += b"\0" * (88 - len(heap_data))
    heap_hdr = 32
    heap_daddr = heap + heap_hdr
    snod = heap_daddr + len(heap_data)
    snod_size = 8 + 2 * K_leaf * 40
    p = snod + snod_size
    ohdr_addr, ohdr_bytes = {}, {}
    for n in names:
        a = arrays[n]
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + struct.pack("<" + "Q" * a.ndim, *a.shape)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)
        msgs = _msg(0x0001, space) + _msg(0x0003, _dtype_msg(a.dtype), flags=1) + _msg(0x0005, fill) + \
            _msg(0x0008, struct.pack("<BBQQ", 3, 1, 0, a.nbytes))  # address patched below
        ohdr_addr[n] = p
        ohdr_bytes[n] = msgs
        p += 16 + len(msgs)
    data_addr = {}
    p = (p + 4095) // 4096 * 4096     # page-align raw data (friendlier to mmap / O_DIRECT readers)
    for n in names:
        data_addr[n] = p
        p += (arrays[n].nbytes + 7) // 8 * 8
    eof = p
    # ---- emit ---------------------------------------------------------------------------------
    with open(path, "wb") as fh:
        sb = SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", K_leaf, K_int, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_ohdr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == sb_size
        fh.write(sb)
        root_msgs = _msg(0x0011, struct.pack("<QQ", btree, heap))
        fh.write(struct.pack("<BBHII4x", 1, 0, 1, 1, len(root_msgs)) + root_msgs)
        last_name = name_off[names[-1]] if names else 0
        bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
        bt += struct.pack("<QQQ", 0, snod, last_name)
        bt += b"\0" * (btree_size - len(bt))
        fh.write(bt)
        fh.write(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_daddr))  # free list: H5HL_FREE_NULL
        fh.write(heap_data)
        sn = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            sn += struct.pack("<QQII16x", name_off[n], ohdr_addr[n], 0, 0)
        sn += b"\0" * (snod_size - len(sn))
        fh.write(sn)
        for n in names:
            a = arrays[n]
            msgs = ohdr_bytes[n]
            # patch the layout address (last message: 8-byte header + version, class, then address)
            lay = _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_addr[n], a.nbytes))
            msgs = msgs[:-len(lay)] + lay
            assert fh.tell() == ohdr_addr[n]
            fh.write(struct.pack("<BBHII4x", 1, 0, 4, 1, len(msgs)) + msgs)
        fh.write(b"\0" * (data_addr[names[0]] - fh.tell()) if names else b"")
        for n in names:
            a = arrays[n]
            assert fh.tell() == data_addr[n]
            fh.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
            fh.write(b"\0" * (-a.nbytes % 8))
        assert fh.tell() == eof


def make_synthetic_cutouts(path, n=4096, channels=5, size=64, seed=1234, nan_fraction=0.0, with_labels=False):
    """Synthetic cutout file in the reference schema: N(0,1) pixels clipped at -3,
    ra ~ U(0,360), dec ~ U(-90,90); ``nan_fraction`` of the (sample, channel) planes set to NaN."""
    rng = np.random.default_rng(seed)
    cut = rng.standard_normal((n, channels, size, size), dtype=np.float32)
    np.maximum(cut, -3.0, out=cut)
    if nan_fraction > 0:
        sel = rng.random((n, channels)) < nan_fraction
        cut[sel] = np.nan
    d = {"cutouts": cut, "ra": rng.uniform(0, 360, n).astype(np.float32),
         "dec": rng.uniform(-90, 90, n).astype(np.float32)}
    if with_labels:
        d["zspec"] = rng.uniform(0, 2, n).astype(np.float32)
        d["zspec_err"] = np.full(n, 0.01, np.float32)
        d["class"] = rng.integers(0, 3, n).astype(np.int64)
    write_datasets(path, d)
    return path
