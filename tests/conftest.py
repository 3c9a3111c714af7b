import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "dealii-slod_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 20250614


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_fields(so, cfg, dist="D100", seed=SEED):
    """Synthetic per-qp coefficient fields (one per spacedim component field)."""
    NE = so.n_cells_per_side(cfg) * cfg.n_sub
    if dist == "const":
        return [np.ones(NE * NE * 4) for _ in range(cfg.spacedim)]
    d, lo, hi = {"D100": (0, 1.0, 100.0), "D1e4": (1, 1.0, 1.0e4)}[dist]
    return [so.fill_coefficient(seed + f, d, lo, hi, NE) for f in range(cfg.spacedim)]


@pytest.fixture(scope="session")
def so():
    import slod_oracle
    slod_oracle.lib()
    return slod_oracle
