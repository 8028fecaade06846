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
    assert ctypes.sizeof(_lib.GemmArgs) == 248 and _lib.GemmArgs.split_k.offset == 224
    assert _lib.GemmArgs.prefetch_wgs.offset == 228 and _lib.GemmArgs.prefetch.offset == 232 and _lib.GemmArgs.prefetch_bytes.offset == 240
    assert _lib.GemmArgs.tile.offset == 192 and _lib.GemmArgs.out_f32.offset == 144 and _lib.GemmArgs.colsum_a.offset == 200


def test_integration_md_bindings_match_the_prototypes():
    """Every `_L.<name>.argtypes = [...]` a maintainer is shown in INTEGRATION.md has the header's parameter count, the
    same pointer / integer / float kind per position, and every call of that name in the document passes as many values."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shown = re.findall(r"_L\.(skyemb_[a-z0-9_]+)\.argtypes\s*=\s*\[([^\]]*)\]", text)
    assert len(shown) >= 5
    kind = {"vp": "p", "i32": "i", "i64": "i", "f32": "f", "f64": "f"}

    def proto_kind(t):
        if t in (ctypes.c_float, ctypes.c_double):
            return "f"
        if t in (ctypes.c_int32, ctypes.c_int64):
            return "i"
        return "p"
    for name, body in shown:
        assert name in _lib.PROTOTYPES, f"INTEGRATION.md binds {name}, which include/skyemb.h does not declare"
        toks = [t.strip() for t in body.split(",") if t.strip()]
        want = _lib.PROTOTYPES[name][1]
        assert len(toks) == len(want), f"INTEGRATION.md: {name}.argtypes has {len(toks)} entries, the header {len(want)}"
        for k, (t, w) in enumerate(zip(toks, want)):
            assert kind[t] == proto_kind(w), f"INTEGRATION.md: {name} argument {k} is shown as {t}"
        # calls of the form _L.name( ... ) inside _chk(...): count top-level commas
        for m in re.finditer(r"_L\." + name + r"\(", text):
            depth, i, n_args, start = 1, m.end(), 1, m.end()
            while depth:
                c = text[i]
                depth += c in "([" 
                depth -= c in ")]"
                n_args += (c == "," and depth == 1)
                i += 1
            if text[start:i - 1].strip():
                assert n_args == len(want), f"INTEGRATION.md: a call of {name} passes {n_args} values, the header takes {len(want)}"


def test_the_product_library_has_no_launch_skipping_switch():
    """skyemb_debug_skip (bench.py's measurement aid: GEMM launches become no-ops) is compiled into libskyemb_measure.so only; in the
    product library the call fails and leaves nothing set.  No compute call: loads and calls two host functions."""
    import ctypes
    import os
    from sky_embeddings_amd import _lib
    here = os.path.dirname(_lib.SO_PATH)
    prod = ctypes.CDLL(os.path.join(here, "libskyemb.so"))
    prod.skyemb_last_error.restype = ctypes.c_char_p
    assert prod.skyemb_debug_skip(1) == -1 and b"product build" in prod.skyemb_last_error()
    assert prod.skyemb_debug_skip(0) == -1
    meas_path = os.path.join(here, "libskyemb_measure.so")
    assert os.path.exists(meas_path), "make -C sky_embeddings_amd/csrc builds the measurement library beside the product one"
    meas = ctypes.CDLL(meas_path)
    assert meas.skyemb_debug_skip(1) == 0 and meas.skyemb_debug_skip(0) == 1
    # same exported surface
    import subprocess
    syms = [set(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", p], text=True).splitlines() if " T skyemb_" in l)
            for p in (os.path.join(here, "libskyemb.so"), meas_path)]
    assert syms[0] == syms[1]
