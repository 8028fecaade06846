"""CPU: the C-ABI library builds, loads and exports every symbol include/skyemb.h declares."""
import ctypes
import os
import re

from sky_embeddings_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "skyemb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skyemb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    so = _lib.build()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/skyemb.h but not exported"
    # the ctypes prototypes cover the header exactly
    assert sorted(_lib.PROTOTYPES) == names


def test_loader_and_error_channel():
    L = _lib.lib()
    assert L.skyemb_version() >= 100
    # argument validation happens before any device work, so it is safe without a GPU
    g = _lib.GemmArgs()
    rc = L.skyemb_gemm(ctypes.byref(g), None)
    assert rc != 0 and b"empty problem" in L.skyemb_last_error()
    rc = L.skyemb_cosine_topk(None, None, None, None, 0, 0, 0, 0, 0.0, 0, 0, None, None, None, None)
    assert rc != 0 and b"bad shape" in L.skyemb_last_error()


def test_gemm_args_struct_layout_matches_header():
    # offsets a C compiler gives the struct (natural alignment) -- guards against ctypes drift
    assert ctypes.sizeof(_lib.GemmArgs) == 232
    assert _lib.GemmArgs.tile.offset == 192 and _lib.GemmArgs.out_f32.offset == 144 and _lib.GemmArgs.colsum_a.offset == 200
