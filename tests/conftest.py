import os
import sys


import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the oracle is test infrastructure only

DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import dpgo_amd
        return dpgo_amd.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must FAIL loudly (no silent skip / fallback); plain runs
    # without -m simply skip gpu tests when there is no device.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no HIP device")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import dpgo_oracle
    return dpgo_oracle


def tiles_to_matrix(Xt):
    """[n, d+1, r] oracle view -> (r, (d+1)n) reference Matrix (same bytes, Fortran order)."""
    n, b, r = Xt.shape
    return np.ascontiguousarray(Xt).reshape(n * b, r).T


def matrix_to_tiles(X, d):
    r, N = X.shape
    return np.ascontiguousarray(np.asfortranarray(X).T).reshape(N // (d + 1), d + 1, r)


def to_product_measurements(om):
    """oracle Measurements -> dpgo_amd.RelativeSEMeasurements"""
    import dpgo_amd
    c = np.copy  # no aliasing: solveRobustPGO updates the product-side weights in place
    return dpgo_amd.RelativeSEMeasurements(om.d, c(om.r1), c(om.p1), c(om.r2), c(om.p2), c(om.R), c(om.t), c(om.kappa),
                                           c(om.tau), c(om.weight), c(om.fixed))


def device_tcg_mode(n, d, r):
    """The tCG arithmetic the device runs (oracle `hess_recurrence` argument): H delta is advanced by the recurrence
    H delta' = beta H delta - H z."""
    return True
