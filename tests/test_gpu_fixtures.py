"""HIP path against the committed golden fixtures (independent numpy/scipy restatement)."""
import numpy as np
import pytest

from fixture_utils import fixtures, global_fields, tolerances
from test_gpu_parity import _mk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", fixtures(), ids=lambda c: c["key"])
def test_gpu_matches_fixture(so, case):
    cfg, g = _mk(so, **case["cfg"])
    fields = global_fields(case)
    for f, a in enumerate(fields):
        g.set_coefficient(f, a)
    basis, premult, offs = g.compute_basis(np.array([case["pid"]], dtype=np.uint32))
    s = cfg.spacedim
    n = case["phi"].size
    tol = tolerances(case)
    a_inf = np.abs(so.assemble_patch(cfg, fields, case["pid"])).sum(axis=(1, 3)).max()
    assert np.abs(basis[:n].reshape(s, -1) - case["phi"]).max() <= tol
    assert np.abs(premult[:n].reshape(s, -1) - case["psi"]).max() <= tol * a_inf
