"""BASELINE config C3 (2-D Poisson H=1/128, n_sub=16, oversampling 3) on its REAL patch shapes.

An 8x8 coarse grid cannot hold C3's dominant shapes: with l=3 every 7x7-cell patch of an 8x8 grid
touches two domain sides.  C3 itself is 91 % fully interior 7x7 patches (448 id-99 boundary rows)
plus one-side 7x7 patches (337 rows).  Both classes exist from a 16x16 grid on, so the parity
tests below run C3's n_sub / oversampling on nref=4 (256 patches) against the oracle, decisions
included; the full-size run checks the size-independent properties on all 16 384 patches of C3.
"""
import numpy as np
import pytest

from conftest import make_fields
from test_gpu_parity import TOL_PHI, _mk, _upload

pytestmark = pytest.mark.gpu


def _morton(x, y):
    r = 0
    for b in range(12):
        r |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return r


def _run_and_diag(g, ids):
    """plan path: outputs on the device, decisions of the selection stage exported"""
    import torch
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    plan = g.plan(ids)
    dev = torch.device("cuda", 0)
    b = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
    q = torch.zeros_like(b)
    plan.execute(b.data_ptr(), q.data_ptr())
    torch.cuda.synchronize()
    plan.status()
    return plan, b.cpu().numpy(), q.cpu().numpy(), plan.diagnostics()


@pytest.mark.parametrize("dist", ["D100", "D1e4"])
def test_c3_real_patch_shapes(so, dist):
    """nref=4, n_sub=16, l=3: fully interior 7x7 patches (N_b = 448), one of every one-side 7x7
    shape (N_b = 337), and one patch of every other (shape, sides) class of the grid, through the
    default kernel at 1e-10 with n_cut / n_dropped / ||delta||_inf equal to the oracle's
    (reference LOD.cc:598-757 at the extents of LOD.cc:140-181)."""
    cfg, g = _mk(so, nref=4, n_sub=16, oversampling=3, stabilize=1)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    classes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        classes.setdefault((i.mx, i.my, tuple(i.side_domain)), []).append(pid)
    interior = classes[(7, 7, (0, 0, 0, 0))]
    assert len(interior) == 64                      # cells 4..11 in both directions
    one_side = [k for k in classes if k[0] == 7 and k[1] == 7 and sum(k[2]) == 1]
    assert len(one_side) == 4
    ids = sorted({v[0] for v in classes.values()} | {v[len(v) // 2] for v in classes.values()}
                 | {_morton(8, 8), _morton(3, 3 + 1), _morton(12, 7), _morton(5, 9), _morton(10, 4)})
    ids = np.array(ids, dtype=np.uint32)
    plan, basis, premult, dg = _run_and_diag(g, ids)
    seen_448 = seen_337 = 0
    worst = worst_psi = 0.0
    widened = []
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid))
        seen_448 += p.n_b == 448
        seen_337 += p.n_b == 337
        phi0, psi0, diag = so.patch_basis(cfg, fields, int(pid))
        spread, stable = so.selection_conditioning(cfg, fields, int(pid))
        off = k * plan.stride
        got, gpsi = basis[off:off + p.n_f], premult[off:off + p.n_f]
        assert np.isfinite(got).all() and np.isfinite(gpsi).all()
        err = np.abs(got - phi0.ravel()).max()
        a_inf = np.abs(so.assemble_patch(cfg, fields, int(pid))).sum(axis=(1, 3)).max()
        epsi = np.abs(gpsi - psi0.ravel()).max() / a_inf
        if stable:
            assert (dg[k].n_cut, dg[k].n_dropped) == (diag.n_cut[0], diag.n_dropped[0]), \
                "patch %d (%dx%d, N_b %d): decisions gpu (%d,%d) oracle (%d,%d)" % (
                    pid, p.mx, p.my, p.n_b, dg[k].n_cut, dg[k].n_dropped, diag.n_cut[0], diag.n_dropped[0])
            assert abs(dg[k].dinf - diag.dinf[0]) <= 1e-8, "patch %d" % pid
        tol = TOL_PHI if (stable and spread <= TOL_PHI) else max(TOL_PHI, 10.0 * spread)
        if tol > TOL_PHI:
            widened.append((int(pid), p.mx, p.my, p.n_b, spread, err))
        assert err <= tol, "patch %d (%dx%d, N_b %d): |dphi| %.3e (tol %.1e)" % (pid, p.mx, p.my, p.n_b, err, tol)
        assert epsi <= tol, "patch %d: |dpsi|/|A| %.3e" % (pid, epsi)
        if p.n_b == 448:
            # C3's dominant class (91 % of its patches) is well conditioned: the 1e-10 bar unwidened
            assert tol == TOL_PHI, "patch %d: oracle spread %.2e" % (pid, spread)
        worst, worst_psi = max(worst, err), max(worst_psi, epsi)
    assert seen_448 >= 4 and seen_337 >= 8
    # widened tolerances only where the oracle itself moves under 1e-13 solver noise: never on interior patches
    assert all(w[3] != 448 for w in widened)
    print("C3 real shapes %s: %d patches (%d with N_b=448, %d with N_b=337), worst |dphi| %.3e, |dpsi|/|A| %.3e; "
          "widened tolerance on %d rim patches: %s" % (dist, len(ids), seen_448, seen_337, worst, worst_psi,
                                                       len(widened), ["%d(%dx%d) %.1e" % (w[0], w[1], w[2], w[4])
                                                                      for w in widened]))


def test_c3_full_size_properties(so):
    """All 16 384 patches of C3 (the configuration bench.py --config C3 times), properties that need
    no oracle: two executions bit-identical; unit l2 norm; phi = 0 on every patch
    boundary dof; psi = 0 on id-0 dofs; internal psi takes at most (2mx-1)(2my-1) distinct |values|
    (the (h^2/4){1,2,4}-weighted cell pattern); every patch reports a decision path; and a sample of
    interior / one-side 7x7 patches agrees with the oracle at 1e-10."""
    import torch
    cfg, g = _mk(so, nref=7, n_sub=16, oversampling=3, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(fields[0]).to(dev)
    g.set_coefficient_device(0, t.data_ptr(), t.numel())
    npatch = g.num_patches
    assert npatch == 16384
    ids = np.arange(npatch, dtype=np.uint32)
    plan = g.plan(ids)
    stride = plan.stride
    assert stride == 113 * 113
    b1 = torch.zeros(npatch * stride, dtype=torch.float64, device=dev)
    p1 = torch.zeros_like(b1)
    plan.execute(b1.data_ptr(), p1.data_ptr())
    torch.cuda.synchronize()
    plan.status()
    dg = plan.diagnostics()
    b2 = torch.zeros_like(b1)
    p2 = torch.zeros_like(b1)
    plan.execute(b2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    plan.status()
    assert torch.equal(b1, b2) and torch.equal(p1, p2)
    del b2, p2
    assert torch.isfinite(b1).all() and torch.isfinite(p1).all()
    nrm = b1.view(npatch, stride).norm(dim=1)
    assert float((nrm - 1.0).abs().max()) < 1e-12
    assert all(d.path in (1, 2) for d in dg)
    full = [pid for pid in range(0, npatch, 61) if g.patch_layout(pid).mx == 7 and g.patch_layout(pid).my == 7]
    assert len(full) > 200
    fb = b1.view(npatch, 113, 113)[full]
    assert float(fb[:, 0, :].abs().max()) == 0 and float(fb[:, -1, :].abs().max()) == 0
    assert float(fb[:, :, 0].abs().max()) == 0 and float(fb[:, :, -1].abs().max()) == 0
    hb, hp = b1.cpu().numpy(), p1.cpu().numpy()
    for pid in range(0, npatch, 331):
        info = g.patch_layout(pid)
        n = info.n_fine
        phi = hb[pid * stride:pid * stride + n].reshape(info.ny + 1, info.nx + 1)
        psi = hp[pid * stride:pid * stride + n].reshape(info.ny + 1, info.nx + 1)
        assert np.all(phi[0, :] == 0) and np.all(phi[-1, :] == 0) and np.all(phi[:, 0] == 0) and np.all(phi[:, -1] == 0)
        sd = list(info.side_domain)
        if sd[0]:
            assert np.all(psi[:, 0] == 0)
        if sd[1]:
            assert np.all(psi[:, -1] == 0)
        if sd[2]:
            assert np.all(psi[0, :] == 0)
        if sd[3]:
            assert np.all(psi[-1, :] == 0)
        inner = psi[1:-1, 1:-1]
        scale = np.abs(inner).max()
        distinct = np.unique(np.round(np.abs(inner) / scale, 6))
        assert distinct.size <= (2 * info.mx - 1) * (2 * info.my - 1) + 1, "patch %d: %d" % (pid, distinct.size)
    # oracle companions on C3's two dominant classes, at full size
    sample = [_morton(64, 64), _morton(17, 90), _morton(3, 50), _morton(124, 77), _morton(40, 3), _morton(99, 124)]
    worst = 0.0
    for pid in sample:
        p = so.patch_info(cfg, pid)
        assert p.mx == 7 and p.my == 7 and p.n_b in (448, 337)
        phi0, psi0, diag = so.patch_basis(cfg, fields, pid)
        err = np.abs(hb[pid * stride:pid * stride + p.n_f] - phi0.ravel()).max()
        assert err <= TOL_PHI, "C3 patch %d: %.3e" % (pid, err)
        assert (dg[pid].n_cut, dg[pid].n_dropped) == (diag.n_cut[0], diag.n_dropped[0])
        worst = max(worst, err)
    print("C3 full size: 16384 patches, bit-reproducible, |norm-1| %.1e, sample worst |dphi| vs oracle %.3e"
          % (float((nrm - 1.0).abs().max()), worst))
