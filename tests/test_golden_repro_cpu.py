"""CPU, build container only: tests/golden/*.npz are exactly what tests/golden/make_golden.py produces from the reference NOW.

Guards against fixture drift (a generator edited after its fixtures were committed, or the other way round).  Skipped where
/root/reference does not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def arrays_equal(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind in "fc":
        return np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "utils")), reason="the reference is not present on this machine")
def test_make_golden_reproduces_the_committed_fixtures(tmp_path):
    env = dict(os.environ, SKYEMB_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    made = sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    kept = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert made == kept, (set(made) ^ set(kept))
    for name in kept:
        new, old = np.load(os.path.join(str(tmp_path), name)), np.load(os.path.join(GOLDEN, name))
        assert sorted(new.files) == sorted(old.files), (name, set(new.files) ^ set(old.files))
        bad = [k for k in old.files if not arrays_equal(new[k], old[k])]
        assert not bad, (name, bad[:10])
