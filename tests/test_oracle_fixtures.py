"""The C oracle against the golden fixtures of the independent numpy/scipy restatement
(literal LOD.cc:345-767 incl. Gram matrix + LAPACK dgesdd), and oracle self-consistency."""
import numpy as np
import pytest

from fixture_utils import fixtures, global_fields, tolerances


@pytest.mark.parametrize("case", fixtures(), ids=lambda c: c["key"])
def test_oracle_matches_fixture(so, case):
    cfg = so.make_cfg(**case["cfg"])
    fields = global_fields(case)
    phi, psi, diag = so.patch_basis(cfg, fields, case["pid"])
    tol = tolerances(case)
    a_inf = np.abs(so.assemble_patch(cfg, fields, case["pid"])).sum(axis=(1, 3)).max()
    assert np.abs(phi - case["phi"]).max() <= tol
    assert np.abs(psi - case["psi"]).max() <= tol * a_inf
    if cfg.stabilize and not so.patch_info(cfg, case["pid"]).is_lod:
        assert list(diag.n_dropped)[:cfg.spacedim] == list(case["n_dropped"])   # same truncation decisions


def test_svd_modes_agree(so):
    """one-sided Jacobi on BD' (default) vs the literal Gram-matrix formulation (mode 1)."""
    from conftest import make_fields
    cfg = so.make_cfg(nref=3, n_sub=4, oversampling=1, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    try:
        for pid in (0, 9, 27, 63):
            so.set_svd_mode(0)
            a, _, da = so.patch_basis(cfg, fields, pid)
            so.set_svd_mode(1)
            b, _, db = so.patch_basis(cfg, fields, pid)
            assert np.abs(a - b).max() < 1e-10
            assert list(da.n_dropped) == list(db.n_dropped)
    finally:
        so.set_svd_mode(0)


def test_openmp_many_equals_single(so):
    from conftest import make_fields
    cfg = so.make_cfg(nref=2, n_sub=4, oversampling=1, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    ids = np.arange(16)
    sizes = np.array([so.patch_info(cfg, int(p)).n_f for p in ids])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    phi, psi = so.basis_many(cfg, fields, ids, offs, int(sizes.sum()), nthreads=4)
    for k, pid in enumerate(ids):
        p1, s1, _ = so.patch_basis(cfg, fields, int(pid))
        assert np.array_equal(phi[offs[k]:offs[k] + sizes[k]], p1.ravel())
        assert np.array_equal(psi[offs[k]:offs[k] + sizes[k]], s1.ravel())
