#!/usr/bin/env python3
"""Fingerprint of the kernel sources a counter profile was taken on: sha256 over csrc/*.hip, *.h, *.cpp (names + contents), first 16
hex digits; feeder.cpp is left out (host-only file I/O: gather threads, HDF5 un-chunking, FITS tile decoding -- no kernel, no launch).  The PMC summaries in profiles/ carry it (`csrc_sha16`); bench.py recomputes it at run time -- there is no .git on the
GPU box -- and flags hardware-counter figures whose kernels have changed since (`pmc_current: false`)."""
import glob, hashlib, os, sys


def csrc_sha16(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "sky_embeddings_amd", "csrc", "*"))):
        if path.endswith((".hip", ".h", ".cpp")) and os.path.basename(path) != "feeder.cpp":
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1:                          # stamp JSON files in place
        import json
        for p in sys.argv[1:]:
            d = json.load(open(p))
            d["csrc_sha16"] = csrc_sha16()
            json.dump(d, open(p, "w"), indent=1)
    print(csrc_sha16())
