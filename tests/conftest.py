import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> list of dict(case) from tests/golden/<name>.npz (data only, allow_pickle stays False)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    n = int(z["n_cases"])
    cases = [dict() for _ in range(n)]
    for key in z.files:
        if key == "n_cases":
            continue
        ci, field = key.split("/", 1)
        cases[int(ci[1:])][field] = z[key]
    return cases


def assert_close(got, ref, rtol=1e-4, atol=1e-5, what=""):
    """fp32 bar of north_star: 1e-4 relative (plus a small absolute floor scaled to the tensor)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s shape %s vs %s" % (what, got.shape, ref.shape)
    scale = max(1.0, float(np.max(np.abs(ref)))) if ref.size else 1.0
    err = np.abs(got - ref)
    tol = atol * scale + rtol * np.abs(ref)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d off; worst at %s got %r ref %r" % (what, bad.sum(), bad.size, i, got[i], ref[i]))


def assert_exact(got, ref, what=""):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, "%s shape %s vs %s" % (what, got.shape, ref.shape)
    if not np.array_equal(got, ref):
        bad = got != ref
        i = np.unravel_index(np.argmax(bad), bad.shape)
        raise AssertionError("%s: %d/%d differ; first at %s got %r ref %r" % (what, bad.sum(), bad.size, i, got[i], ref[i]))
