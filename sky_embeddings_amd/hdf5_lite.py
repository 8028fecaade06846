"""Minimal HDF5 access for the cutout files of the hot path (no h5py on the build / GPU boxes).

Schema (reference ``data_processing/utils.py:333-361``, ``configs/README.md:10-13``): root-level
datasets ``cutouts float32 [N,C,H,W]``, ``ra``/``dec`` float32 [N], optional ``zspec``,
``zspec_err``, ``class``.  The per-tile temp files and the train/validation splits are created with
``create_dataset(name, shape, dtype)`` (CONTIGUOUS layout: data_processing/utils.py:346-350,
4_split_dataset.py:32); the combined files with ``create_dataset(k, shape, maxshape=(None, ...))`` +
``resize`` (CHUNKED layout with h5py's automatic chunk shape, no filters:
data_processing/2_create_h5_files.py:70-81, 3_combine_h5_files.py:37-52, create_datasets.py:78-85).
This module reads that subset of the HDF5 1.x file format (superblock v0/v1, v1 object headers incl.
continuation blocks, v1 group B-trees + local heaps, dataspace v1/v2 incl. maximum dimensions,
fixed-point / IEEE float little-endian datatypes, data layout v3: contiguous, compact and chunked with the
v1 chunk B-tree) and writes it (the synthetic-data generator; chunked files for round-trip tests).
Chunked datasets may be filtered with what h5py offers without plugins (``compression='gzip'``, ``shuffle``, ``fletcher32``);
other filters and new-style-group files raise ``NotImplementedError`` naming the feature.

Contiguous datasets are exposed as ``numpy.memmap`` views, so a [N,5,64,64] cutout file is read with plain
coalesced page-cache I/O instead of the reference's per-item ``h5py.File`` open (utils/dataloaders.py:289).
A chunked dataset answers indexed reads by assembling the chunks it touches; the feeder's fast path
(``Dataset._array()``) un-chunks it ONCE (native threads, ``skyemb_h5_unchunk_host``) into a contiguous
cache file next to the source (or under ``$SKYEMB_H5_CACHE``) and memory-maps that.

If ``h5py`` is importable it is NOT used: one code path everywhere keeps results reproducible.
"""
from __future__ import annotations

import json
import os
import struct
import tempfile

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5LiteError(IOError):
    pass


# ------------------------------------------------------------------------------------------ reader
class Dataset:
    def __init__(self, name, shape, dtype, offset, fileobj_path, inline=None, chunks=None, btree=None, reader=None, filters=None):
        self.name, self.shape, self.dtype = name, tuple(shape), np.dtype(dtype)
        self.filters = list(filters or [])                 # [(filter id, client data)] in the order the writer applied them
        self._offset, self._path, self._inline = offset, fileobj_path, inline
        self.chunks = tuple(chunks) if chunks is not None else None      # None: contiguous / compact
        self._btree, self._reader = btree, reader
        self._mm = None
        self._table = None

    # -- chunked layout ---------------------------------------------------------------------------------------------
    def chunk_table(self):
        """(addresses int64 [n], element offsets int64 [n, rank]) of every stored chunk (v1 B-tree, node type 1)."""
        if self._table is None:
            addr, off, sizes = [], [], []
            if self._btree not in (None, UNDEF):
                self._reader._walk_chunk_btree(self._btree, len(self.shape), int(np.prod(self.chunks)) * self.dtype.itemsize,
                                               addr, off, sizes if self.filters else None)
            self._table = (np.asarray(addr, dtype=np.int64), np.asarray(off, dtype=np.int64).reshape(len(addr), len(self.shape)))
            self._stored = sizes                                # filtered datasets: (stored bytes, filter mask) per chunk
        return self._table

    def _decode_chunk(self, raw, fmask):
        """A stored chunk through the dataset's filter pipeline in reverse (HDF5 file format, "Filter Pipeline" message): the
        filters h5py offers without plugins -- deflate (1, ``compression='gzip'``), shuffle (2), fletcher32 (3; the trailing
        checksum is dropped, not verified).  Bit i of the chunk's filter mask set = filter i was skipped for this chunk."""
        import zlib
        for i in range(len(self.filters) - 1, -1, -1):
            if (fmask >> i) & 1:
                continue
            fid = self.filters[i][0]
            if fid == 3:
                raw = raw[:-4]
            elif fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = self.dtype.itemsize
                n = len(raw) // es
                body = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n).T.tobytes()
                raw = body + raw[n * es:]
        want = int(np.prod(self.chunks)) * self.dtype.itemsize
        if len(raw) != want:
            raise H5LiteError(f"{self._path}:{self.name}: a chunk decodes to {len(raw)} bytes, expected {want}")
        return raw

    def _unchunk_filtered(self, out):
        """Every stored chunk decoded (threads: zlib releases the interpreter lock) and scattered into the contiguous image."""
        from concurrent.futures import ThreadPoolExecutor
        addr, off = self.chunk_table()
        src = np.memmap(self._path, dtype=np.uint8, mode="r")
        rank = len(self.shape)

        def one(ci):
            size, fmask = self._stored[ci]
            a = int(addr[ci])
            chunk = np.frombuffer(self._decode_chunk(bytes(src[a:a + size]), fmask), dtype=self.dtype).reshape(self.chunks)
            o = off[ci]
            ext = [min(self.chunks[d], self.shape[d] - int(o[d])) for d in range(rank)]
            out[tuple(slice(int(o[d]), int(o[d]) + ext[d]) for d in range(rank))] = chunk[tuple(slice(0, e) for e in ext)]

        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            for _ in pool.map(one, range(len(addr)), chunksize=64):
                pass

    def _cache_path(self):
        root = os.environ.get("SKYEMB_H5_CACHE")
        base = f".{os.path.basename(self._path)}.{self.name}.contig"
        home = os.path.dirname(os.path.abspath(self._path))
        for d in ([root] if root else []) + [home, tempfile.gettempdir()]:
            if d and os.path.isdir(d) and os.access(d, os.W_OK):
                if d != root and d != home:
                    # a dataset-sized file in the temp directory is rarely what the operator wants: say so once per dataset
                    import warnings
                    warnings.warn(f"hdf5_lite: {home} is read-only, the contiguous copy of {self._path}:{self.name} goes to {d} "
                                  f"(set SKYEMB_H5_CACHE to choose the directory)", RuntimeWarning, stacklevel=3)
                return os.path.join(d, base)
        raise H5LiteError(f"no writable directory for the contiguous cache of {self._path}:{self.name}")

    def _cache_valid(self, path, meta, stamp, nbytes):
        try:
            if os.path.getsize(path) != nbytes:
                return False
            with open(meta) as fh:
                return json.load(fh) == stamp
        except (OSError, ValueError):
            return False

    def _unchunk(self):
        """Contiguous row-major copy of a chunked dataset in a cache file (rebuilt when the source changes).  Every rank's
        feeder and every loader worker lands here at the same moment: ONE process builds the copy under an exclusive
        lock (flock on `<cache>.lock`), the others block on the lock and then find the finished file.  The stamp (.json) is
        renamed into place BEFORE the data file, so a reader that sees the data file also sees the stamp that describes it."""
        import fcntl
        from ._lib import check, lib
        st = os.stat(self._path)
        stamp = {"size": st.st_size, "mtime_ns": st.st_mtime_ns, "shape": list(self.shape), "dtype": self.dtype.str,
                 "chunks": list(self.chunks)}
        path = self._cache_path()
        meta = path + ".json"
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        if self._cache_valid(path, meta, stamp, nbytes):
            return np.memmap(path, dtype=self.dtype, mode="r", shape=self.shape)
        with open(path + ".lock", "w") as lock:
            locked = True
            try:
                fcntl.flock(lock, fcntl.LOCK_EX)                # released when the file is closed (also if the builder dies)
            except OSError:
                # some NFS / Lustre mounts have no flock: build unlocked.  Still safe -- every builder writes its own
                # <pid>.tmp and renames it into place atomically; concurrent builders only duplicate the work.
                locked = False
            try:
                if not self._cache_valid(path, meta, stamp, nbytes):      # somebody else may have built it while we waited
                    addr, off = self.chunk_table()
                    tmp = f"{path}.{os.getpid()}.tmp"
                    try:
                        out = np.memmap(tmp, dtype=self.dtype, mode="w+", shape=self.shape)   # holes (never-written chunks) read as 0
                        src = np.memmap(self._path, dtype=np.uint8, mode="r")
                        cd = np.asarray(self.chunks, dtype=np.int64)
                        dd = np.asarray(self.shape, dtype=np.int64)
                        if self.filters:
                            self._unchunk_filtered(out)
                        else:
                            check(lib().skyemb_h5_unchunk_host(src.ctypes.data, src.size, addr.ctypes.data,
                                                               np.ascontiguousarray(off).ctypes.data, len(addr), len(self.shape),
                                                               cd.ctypes.data, dd.ctypes.data, self.dtype.itemsize, out.ctypes.data,
                                                               min(16, os.cpu_count() or 1)), "skyemb_h5_unchunk_host")
                        out.flush()
                        del out, src
                        if os.path.exists(path):
                            os.unlink(path)                     # a stale copy must not be seen next to the new stamp
                        with open(meta + ".tmp", "w") as fh:
                            json.dump(stamp, fh)
                        os.replace(meta + ".tmp", meta)
                        os.replace(tmp, path)
                    finally:
                        if os.path.exists(tmp):
                            os.unlink(tmp)
            finally:
                if locked:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        return np.memmap(path, dtype=self.dtype, mode="r", shape=self.shape)

    def _read_chunked(self, idx):
        """Indexed read of a chunked dataset without the cache: only the chunks the request touches are read.  ``idx``
        selects along axis 0 (int, slice or integer array); trailing axes are returned whole."""
        n = self.shape[0]
        if isinstance(idx, tuple):
            return self._read_chunked(idx[0])[(slice(None),) + tuple(idx[1:])] if not isinstance(idx[0], (int, np.integer)) \
                else self._read_chunked(idx[0])[tuple(idx[1:])]
        scalar = isinstance(idx, (int, np.integer))
        rows = np.arange(n)[idx] if not scalar else np.asarray([idx + n if idx < 0 else idx])
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        if rows.size and (rows.min() < 0 or rows.max() >= n):
            raise IndexError(f"index out of range for axis 0 with size {n}")
        out = np.zeros((rows.size,) + self.shape[1:], self.dtype)
        addr, off = self.chunk_table()
        if rows.size and len(addr):
            c0 = self.chunks[0]
            want = {}                                           # axis-0 chunk block -> [(position in out, row inside block)]
            for pos, r in enumerate(rows.tolist()):
                want.setdefault(r // c0, []).append((pos, r % c0))
            sel = np.nonzero(np.isin(off[:, 0] // c0, list(want)))[0]
            csize = int(np.prod(self.chunks)) * self.dtype.itemsize
            with open(self._path, "rb") as fh:
                for ci in sel.tolist():
                    fh.seek(int(addr[ci]))
                    if self.filters:
                        size, fmask = self._stored[ci]
                        chunk = np.frombuffer(self._decode_chunk(fh.read(size), fmask), dtype=self.dtype).reshape(self.chunks)
                    else:
                        chunk = np.frombuffer(fh.read(csize), dtype=self.dtype).reshape(self.chunks)
                    o = off[ci]
                    ext = [min(self.chunks[d], self.shape[d] - int(o[d])) for d in range(1, len(self.shape))]
                    dst_sl = tuple(slice(int(o[d]), int(o[d]) + ext[d - 1]) for d in range(1, len(self.shape)))
                    src_sl = tuple(slice(0, e) for e in ext)
                    for pos, rin in want[int(o[0]) // c0]:
                        out[(pos,) + dst_sl] = chunk[(rin,) + src_sl]
        return out[0] if scalar else out

    # -- array access -----------------------------------------------------------------------------------------------
    def _array(self):
        if self._mm is None:
            if self._inline is not None:
                self._mm = np.frombuffer(self._inline, dtype=self.dtype).reshape(self.shape)
            elif self.chunks is not None:
                self._mm = self._unchunk() if int(np.prod(self.shape)) else np.zeros(self.shape, self.dtype)
            elif self._offset == UNDEF or int(np.prod(self.shape)) == 0:
                self._mm = np.zeros(self.shape, self.dtype)  # never written: fill value 0
            else:
                self._mm = np.memmap(self._path, dtype=self.dtype, mode="r", offset=self._offset, shape=self.shape)
        return self._mm

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        # small 1-D chunked datasets (ra, dec, labels) are assembled from their chunks; image-like ones go through the
        # contiguous cache: h5py's automatic chunks hold ~128 rows x a sliver of the bands / pixels, so ONE cutout touches
        # ~160 chunks (10 MB) -- per-item chunk reads would be hundreds of times the useful bytes
        if self.chunks is not None and self._mm is None and len(self.shape) == 1:
            return self._read_chunked(idx)
        return np.array(self._array()[idx])  # copy, like h5py (callers mutate: utils/dataloaders.py:294)

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._array()) if self.chunks is None or self._mm is not None else self._read_chunked(slice(None))
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
        self._far = None
        self._parse()

    # -- low level
    def _read(self, off, n):
        if off + n > len(self._buf):
            if self._far is None:                  # chunk B-tree nodes are spread over the whole file: map it once
                self._far = np.memmap(self.path, dtype=np.uint8, mode="r")
            return self._far[off:off + n].tobytes()
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

    def _walk_chunk_btree(self, addr, rank, chunk_bytes, out_addr, out_off, out_stored=None):
        """v1 B-tree, node type 1 (raw data chunks): key = {chunk size u32, filter mask u32, rank+1 offsets u64}."""
        if self._read(addr, 4) != b"TREE":
            raise H5LiteError("bad chunk B-tree signature")
        ntype, level, used = self._u(addr + 4, 1), self._u(addr + 5, 1), self._u(addr + 6, 2)
        if ntype != 1:
            raise H5LiteError("unexpected B-tree node type for a chunked dataset")
        ksize = 8 + 8 * (rank + 1)
        node = self._read(addr + 24, used * (ksize + 8) + ksize)
        for i in range(used):
            p = i * (ksize + 8)
            size, fmask = struct.unpack("<II", node[p:p + 8])
            offs = struct.unpack("<" + "Q" * (rank + 1), node[p + 8:p + ksize])
            child = struct.unpack("<Q", node[p + ksize:p + ksize + 8])[0]
            if level > 0:
                self._walk_chunk_btree(child + self._base, rank, chunk_bytes, out_addr, out_off, out_stored)
            else:
                if out_stored is not None:
                    out_stored.append((size, fmask))
                elif fmask != 0 or size != chunk_bytes:
                    raise H5LiteError("a chunk of a dataset without a filter pipeline is stored filtered")
                out_addr.append(child + self._base)
                out_off.append(offs[:rank])

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

    @staticmethod
    def _filter_pipeline(name, d):
        """Filter pipeline message (0x000B), versions 1 and 2 -> [(filter id, client data values)].  What h5py writes without
        plugins is read: deflate (1), shuffle (2), fletcher32 (3); szip, n-bit, scale-offset and registered third-party filters
        (LZF 32000, Blosc 32001, ...) are refused by name."""
        names = {1: "deflate", 2: "shuffle", 3: "fletcher32", 4: "szip", 5: "nbit", 6: "scaleoffset", 32000: "lzf", 32001: "blosc"}
        ver, nf = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid = struct.unpack("<H", d[p:p + 2])[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack("<H", d[p + 2:p + 4])[0]
                p += 4
            else:
                nlen = 0
                p += 2
            _flags, ncd = struct.unpack("<HH", d[p:p + 4])
            p += 4
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = struct.unpack("<" + "I" * ncd, d[p:p + 4 * ncd])
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            if fid not in (1, 2, 3):
                raise NotImplementedError(f"dataset {name!r}: filter {names.get(fid, fid)} (id {fid}) is not supported "
                                          f"(deflate / shuffle / fletcher32 are)")
            out.append((fid, cd))
        return out

    def _dataset(self, name, addr):
        shape = dtype = None
        layout = None
        filters = []
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
                elif cls == 2:
                    nd = d[2]                                  # dimensionality = rank + 1 (last: element size)
                    bt = struct.unpack("<Q", d[3:11])[0]
                    dims = struct.unpack("<" + "I" * nd, d[11:11 + 4 * nd])
                    layout = ("chunked", bt, dims)
                else:
                    raise NotImplementedError(f"dataset {name!r}: data layout class {cls} is not supported")
            elif mtype == 0x000B:
                filters = self._filter_pipeline(name, d)
        if shape is None or layout is None or dtype is None:
            return None  # a sub-group or an unsupported type: not part of the schema
        if layout[0] == "compact":
            return Dataset(name, shape, dtype, None, self.path, inline=layout[1])
        if layout[0] == "chunked":
            bt, dims = layout[1], layout[2]
            if len(dims) != len(shape) + 1 or dims[-1] != np.dtype(dtype).itemsize:
                raise H5LiteError(f"dataset {name!r}: inconsistent chunk dimensions {dims} for shape {shape}")
            return Dataset(name, shape, dtype, None, self.path, chunks=dims[:-1], btree=bt if bt == UNDEF else bt + self._base,
                           reader=self, filters=filters)
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


CHUNK_K = 32          # "indexed storage internal node K" a version-0 superblock implies: <= 64 entries per chunk B-tree node


def _chunk_blobs(a, cs):
    """(element offsets, bytes) of every chunk of array ``a`` cut into chunks of shape ``cs`` (C order of the chunk
    grid; edge chunks are stored whole, zero padded -- as the HDF5 library does)."""
    grid = [-(-a.shape[d] // cs[d]) for d in range(a.ndim)]
    out = []
    for flat in range(int(np.prod(grid))):
        g = np.unravel_index(flat, grid)
        off = [int(g[d]) * cs[d] for d in range(a.ndim)]
        blk = np.zeros(cs, a.dtype)
        sl = tuple(slice(off[d], min(off[d] + cs[d], a.shape[d])) for d in range(a.ndim))
        blk[tuple(slice(0, s_.stop - s_.start) for s_ in sl)] = a[sl]
        out.append((off, blk.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()))
    return out


def _chunk_btree(entries, rank, chunk_bytes, alloc):
    """v1 B-tree over ``entries`` = [(offsets, address)]; ``alloc(nbytes) -> address`` reserves file space.  Returns
    (root address, [(address, node bytes)])."""
    ksize = 8 + 8 * (rank + 1)
    node_bytes = 24 + (2 * CHUNK_K + 1) * ksize + 2 * CHUNK_K * 8

    def key(offsets):
        return struct.pack("<II", chunk_bytes, 0) + struct.pack("<" + "Q" * (rank + 1), *offsets, 0)
    nodes = []
    level = 0
    items = [(off, addr) for off, addr in entries]            # (first key offsets, child address)
    last_off = entries[-1][0]
    while True:
        groups = [items[i:i + 2 * CHUNK_K] for i in range(0, len(items), 2 * CHUNK_K)]
        addrs = [alloc(node_bytes) for _ in groups]
        nxt = []
        for gi, grp in enumerate(groups):
            left = addrs[gi - 1] if gi > 0 else UNDEF
            right = addrs[gi + 1] if gi + 1 < len(groups) else UNDEF
            b = b"TREE" + struct.pack("<BBHQQ", 1, level, len(grp), left, right)
            for off, child in grp:
                b += key(off) + struct.pack("<Q", child)
            b += key(groups[gi + 1][0][0] if gi + 1 < len(groups) else last_off)     # final key: the next node's first chunk
            b += b"\0" * (node_bytes - len(b))
            nodes.append((addrs[gi], b))
            nxt.append((grp[0][0], addrs[gi]))
        if len(groups) == 1:
            return addrs[0], nodes
        items, level = nxt, level + 1


def write_datasets(path, datasets: dict, chunks: dict | None = None):
    """Create an HDF5 file with root-level datasets (h5py-readable, 'earliest' format).  ``datasets`` maps name ->
    ndarray (written little-endian, C order).  Names listed in ``chunks`` (name -> chunk shape) are written CHUNKED and
    resizable along axis 0 -- the layout ``create_dataset(k, shape, maxshape=(None, ...))`` + ``resize`` produces in the
    reference's ETL (data_processing/2_create_h5_files.py:70-81); the others contiguous."""
    chunks = dict(chunks or {})
    names = sorted(datasets)
    if len(names) > 8:
        raise ValueError("hdf5_lite writes at most 8 root datasets (one symbol-table node)")
    arrays = {n: np.ascontiguousarray(datasets[n]) for n in names}
    for n, cs in chunks.items():
        if n not in arrays or len(cs) != arrays[n].ndim or min(cs) < 1:
            raise ValueError(f"bad chunk shape {cs} for dataset {n!r}")
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

    def layout_msg(n, addr):
        a = arrays[n]
        if n in chunks:
            body = struct.pack("<BBBQ", 3, 2, a.ndim + 1, addr) + struct.pack("<" + "I" * (a.ndim + 1), *chunks[n], a.dtype.itemsize)
            return _msg(0x0008, body)
        return _msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, a.nbytes))

    def header_msgs(n, addr):
        a = arrays[n]
        if n in chunks:      # resizable along axis 0: dataspace v1 with maximum dimensions (flags bit 0)
            space = struct.pack("<BBB5x", 1, a.ndim, 1) + struct.pack("<" + "Q" * a.ndim, *a.shape) + \
                struct.pack("<" + "Q" * a.ndim, UNDEF, *a.shape[1:])
        else:
            space = struct.pack("<BBB5x", 1, a.ndim, 0) + struct.pack("<" + "Q" * a.ndim, *a.shape)
        fill = struct.pack("<BBBB", 2, 3 if n in chunks else 2, 0, 0)
        return _msg(0x0001, space) + _msg(0x0003, _dtype_msg(a.dtype), flags=1) + _msg(0x0005, fill) + layout_msg(n, addr)
    ohdr_addr = {}
    for n in names:
        ohdr_addr[n] = p
        p += 16 + len(header_msgs(n, 0))
    data_addr, chunk_plan = {}, {}
    p = (p + 4095) // 4096 * 4096     # page-align raw data (friendlier to mmap / O_DIRECT readers)
    data_start = p
    blobs = []                        # (address, bytes) in file order

    def alloc(nbytes):
        nonlocal p
        a_ = p
        p += (nbytes + 7) // 8 * 8
        return a_
    for n in names:
        a = arrays[n]
        if n in chunks:
            if a.size == 0:
                data_addr[n] = UNDEF
                continue
            cs = tuple(int(c) for c in chunks[n])
            chunk_bytes = int(np.prod(cs)) * a.dtype.itemsize
            entries = []
            for off, raw in _chunk_blobs(a, cs):
                addr = alloc(chunk_bytes)
                blobs.append((addr, raw))
                entries.append((off, addr))
            root, nodes = _chunk_btree(entries, a.ndim, chunk_bytes, alloc)
            blobs.extend(nodes)
            data_addr[n] = root
        else:
            data_addr[n] = alloc(a.nbytes)
            blobs.append((data_addr[n], a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()))
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
            msgs = header_msgs(n, data_addr[n])
            assert fh.tell() == ohdr_addr[n]
            fh.write(struct.pack("<BBHII4x", 1, 0, 4, 1, len(msgs)) + msgs)
        fh.write(b"\0" * (data_start - fh.tell()))
        for addr, raw in sorted(blobs, key=lambda t: t[0]):
            fh.write(b"\0" * (addr - fh.tell()))
            assert fh.tell() == addr
            fh.write(raw)
        fh.write(b"\0" * (eof - fh.tell()))
        assert fh.tell() == eof


def h5py_guess_chunk(shape, itemsize):
    """The chunk shape h5py picks for ``create_dataset(shape=(0, ...), maxshape=(None, ...))`` without ``chunks=`` (its
    ``filters.guess_chunk``: unlimited axes count as 1024, target 16 KiB * 2**log10(size / 1 MiB) clipped to
    [8 KiB, 1 MiB], axes halved in turn until the chunk is within 50 % of the target)."""
    import math
    chunks = [1024 if i == 0 else max(int(x), 1) for i, x in enumerate(shape)]
    dset_size = float(np.prod(chunks)) * itemsize
    target = 16384.0 * (2 ** math.log10(dset_size / (1024.0 * 1024)))
    target = min(max(target, 8192.0), 1048576.0)
    idx = 0
    while True:
        nbytes = float(np.prod(chunks)) * itemsize
        if (nbytes < target or abs(nbytes - target) / target < 0.5) and nbytes < 1048576.0:
            break
        if int(np.prod(chunks)) == 1:
            break
        chunks[idx % len(chunks)] = int(math.ceil(chunks[idx % len(chunks)] / 2.0))
        idx += 1
    return tuple(int(c) for c in chunks)


def make_synthetic_cutouts(path, n=4096, channels=5, size=64, seed=1234, nan_fraction=0.0, with_labels=False, chunked=False):
    """Synthetic cutout file in the reference schema: N(0,1) pixels clipped at -3,
    ra ~ U(0,360), dec ~ U(-90,90); ``nan_fraction`` of the (sample, channel) planes set to NaN.  ``chunked=True`` writes
    every dataset resizable + chunked with h5py's automatic chunk shapes, like the reference's combined files."""
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
    write_datasets(path, d, chunks={k: h5py_guess_chunk((0,) + v.shape[1:], v.dtype.itemsize) for k, v in d.items()} if chunked else None)
    return path
