"""GPU parity for the vector-valued (elasticity) path, the reference quirk switches, the
non-2^k golden geometry and ragged / edge-case plans.  Same tolerances as test_gpu_parity."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_fields
from test_gpu_parity import TOL_PHI, _check_patch, _mk, _upload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stabilize", [0, 1])
@pytest.mark.parametrize("proj_quirk", [0, 1])
def test_elasticity_small_all_patches(so, stabilize, proj_quirk):
    """ElasticityProblem geometry (Elasticity.h): 2 components, H=1/4, n=4, l=1, (lambda,mu) random."""
    cfg, g = _mk(so, nref=2, n_sub=4, oversampling=1, spacedim=2, stabilize=stabilize, proj_quirk=proj_quirk)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    ids = np.arange(g.num_patches)
    basis, premult, offs = g.compute_basis(ids)
    for k, pid in enumerate(ids):
        _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "elast")


@pytest.mark.parametrize("dist", ["D100", "D1e4"])
def test_elasticity_c4_sample(so, dist):
    """BASELINE config C4: 2-D elasticity H=1/32, n=8, l=2 (3362 dofs, 50 candidates per full
    patch).  The oracle needs ~0.1 s per patch: every patch shape (first and middle patch of each)
    plus every 8th patch of the configuration, at both contrasts."""
    cfg, g = _mk(so, nref=5, n_sub=8, oversampling=2, spacedim=2, stabilize=1)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    shapes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), []).append(pid)
    ids = sorted({v[0] for v in shapes.values()} | {v[len(v) // 2] for v in shapes.values()} | {341, 682, 1023}
                 | set(range(0, g.num_patches, 8)))
    ids = np.array(ids)
    basis, premult, offs = g.compute_basis(ids)
    worst = 0.0
    for k, pid in enumerate(ids):
        e = _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "C4")
        worst = max(worst, e[0])
    print("C4 sample of %d patches: worst |dphi| %.3e" % (len(ids), worst))


def test_constant_coefficient_reuse_quirk(so):
    """quirk Q1 (LOD.cc:354-362): with constant_coefficients every full patch re-uses the FIRST
    full patch's matrix even though the field is random."""
    cfg, g = _mk(so, nref=3, n_sub=4, oversampling=1, stabilize=1, reuse_full=1)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    ids = np.arange(g.num_patches)
    basis, premult, offs = g.compute_basis(ids)
    for k, pid in enumerate(ids):
        _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "Q1")
    # full patches with the same boundary ids must be bit-identical now
    groups = {}
    for k in ids:
        i = g.patch_layout(int(k))
        if i.mx == 3 and i.my == 3:
            groups.setdefault(tuple(i.side_domain), []).append(int(k))
    assert len(groups[(0, 0, 0, 0)]) >= 4
    for members in groups.values():
        n = g.patch_layout(members[0]).n_fine
        ref = basis[int(offs[members[0]]):int(offs[members[0]]) + n]
        for k in members[1:]:
            assert np.array_equal(basis[int(offs[k]):int(offs[k]) + n], ref)


def test_solve_poisson_problem_on_patch_01_golden(so):
    """The reference golden tests/solve_poisson_problem_on_patch_01.output through the HIP path:
    10x10 grid (not 2^k), 7 subdivisions, overlap 3, patch of cell (1,4); the sum of the columns
    of X = A_c^{-1} P^T solves -lap u = 1 with zero Dirichlet data (sum_k P^T[:,k] = int phi_i)."""
    gold = np.array([float(x) for x in open(os.path.join(
        GOLDEN, "reference", "solve_poisson_problem_on_patch_01.output")).read().split()])
    cfg, g = _mk(so, n_cells=10, n_sub=7, oversampling=3, stabilize=0)
    g.set_coefficient(0, np.ones(70 * 70 * 4))
    pid = 1 + 4 * 10
    info = g.patch_layout(pid)
    assert (info.x0, info.y0, info.mx, info.my) == (0, 1, 5, 7)
    u = g.patch_solution(pid).sum(axis=1)
    out = np.zeros(71 * 71)
    ix = np.arange(info.n_fine) % (info.nx + 1)
    iy = np.arange(info.n_fine) // (info.nx + 1)
    out[(info.x0 * 7 + ix) + (info.y0 * 7 + iy) * 71] = u
    nz = gold != 0
    assert np.array_equal(nz, out != 0)
    assert np.max(np.abs(out[nz] - gold[nz]) / np.abs(gold[nz])) < 6e-4   # 4 printed digits


def test_ragged_offsets_empty_and_repeated_ids(so):
    cfg, g = _mk(so, nref=3, n_sub=4, oversampling=1, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    # empty plan
    b, p, o = g.compute_basis(np.zeros(0, dtype=np.uint32))
    assert b.size == 0
    # reversed order with gaps between patches, one id twice
    ids = np.array([63, 0, 27, 27, 9], dtype=np.uint32)
    sizes = [g.patch_layout(int(i)).n_fine for i in ids]
    offs = np.cumsum([0] + [s + 7 for s in sizes[:-1]]).astype(np.uint64)
    total = int(offs[-1]) + sizes[-1]
    basis, premult, _ = g.compute_basis(ids, offsets=offs, total=total)
    for k, pid in enumerate(ids):
        _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "ragged")
    for k in range(len(ids) - 1):   # the gaps stay untouched
        gap = basis[int(offs[k]) + sizes[k]:int(offs[k + 1])]
        assert np.all(gap == 0.0)
    # id out of range
    import slod_amd
    with pytest.raises(slod_amd.SlodError) as e:
        g.compute_basis(np.array([64], dtype=np.uint32))
    assert e.value.code == -1


def test_ensemble_of_problems(so):
    """n_problems > 1: gid = problem*num_patches + patch_id picks the coefficient realisation."""
    import slod_amd
    kw = dict(nref=2, n_sub=4, oversampling=1, stabilize=1)
    cfg = so.make_cfg(**kw)
    g = slod_amd.Slod(n_problems=3, **kw)
    NE = g.NE
    fields = [so.fill_coefficient(100 + pb, 0, 1.0, 100.0, NE) for pb in range(3)]
    for pb in range(3):
        g.set_coefficient(0, fields[pb], problem=pb)
    gids = np.array([2 * 16 + 5, 0 * 16 + 5, 1 * 16 + 15], dtype=np.uint32)
    basis, premult, offs = g.compute_basis(gids)
    for k, gid in enumerate(gids):
        _check_patch(so, cfg, [fields[int(gid) // 16]], int(gid) % 16, basis, premult, int(offs[k]), "ens")


def test_size_independent_properties_at_full_size(so):
    """C2 at full size, properties that need no oracle: phi vanishes on every patch
    boundary dof, has unit l2 norm; psi = 0 on id-0 dofs; on internal dofs psi is the (h^2/4){1,2,4}
    weighted cell pattern P^T c / ||phi_raw|| (one value per cell, interior cell edge and interior cell
    vertex: at most (2mx-1)(2my-1) distinct |values| per patch); two executions give bit-identical results."""
    cfg, g = _mk(so, nref=5, n_sub=8, oversampling=2, stabilize=1)
    fields = make_fields(so, cfg, "D1e4")
    _upload(g, fields)
    ids = np.arange(g.num_patches)
    b1, p1, offs = g.compute_basis(ids)
    b2, p2, _ = g.compute_basis(ids)
    assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
    for k in ids[::37]:
        info = g.patch_layout(int(k))
        n = info.n_fine
        phi = b1[int(offs[k]):int(offs[k]) + n].reshape(info.ny + 1, info.nx + 1)
        psi = p1[int(offs[k]):int(offs[k]) + n].reshape(info.ny + 1, info.nx + 1)
        assert abs(np.linalg.norm(phi) - 1.0) < 1e-13
        assert np.all(phi[0, :] == 0) and np.all(phi[-1, :] == 0) and np.all(phi[:, 0] == 0) and np.all(phi[:, -1] == 0)
        sd = list(info.side_domain)
        if sd[0]:
            assert np.all(psi[:, 0] == 0)
        if sd[3]:
            assert np.all(psi[-1, :] == 0)
        inner = psi[1:-1, 1:-1]
        scale = np.abs(inner).max()
        distinct = np.unique(np.round(np.abs(inner) / scale, 7))
        assert distinct.size <= (2 * info.mx - 1) * (2 * info.my - 1) + 1


@pytest.mark.parametrize("mode", ["mf", "tw", "ws", "coop", "nd"])
@pytest.mark.parametrize("spacedim", [1, 2])
def test_all_solver_kernels(so, mode, spacedim, monkeypatch):
    """The patch-solve kernel families (twisted wave-specialised = default, MFMA-factorised,
    wave-specialised, cooperative, nested dissection -- scalar problems only, vector plans fall back
    to the default) must all meet the parity bar; SLOD_SOLVE selects one per plan."""
    monkeypatch.setenv("SLOD_SOLVE", mode)
    kw = dict(nref=3, n_sub=4, oversampling=1, stabilize=1) if spacedim == 1 else \
        dict(nref=2, n_sub=4, oversampling=1, spacedim=2, stabilize=1)
    cfg, g = _mk(so, **kw)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    ids = np.arange(g.num_patches)
    basis, premult, offs = g.compute_basis(ids)
    for k, pid in enumerate(ids):
        _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), mode)


@pytest.mark.parametrize("kw", [dict(nref=1, n_sub=2, oversampling=1), dict(nref=2, n_sub=3, oversampling=2),
                                dict(nref=3, n_sub=2, oversampling=1), dict(nref=2, n_sub=5, oversampling=1),
                                dict(nref=3, n_sub=6, oversampling=1)])
def test_odd_and_tiny_shapes(so, kw):
    """Very small patches (1-3 interior lines), odd n_sub, patches covering the whole domain."""
    cfg, g = _mk(so, stabilize=1, **kw)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    ids = np.arange(g.num_patches)
    basis, premult, offs = g.compute_basis(ids)
    for k, pid in enumerate(ids):
        _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "shape")


@pytest.mark.gpu
@pytest.mark.parametrize("kw,what", [
    (dict(nref=3, n_sub=16, oversampling=1), "192 boundary rows: chunked QR, R carried between chunks"),
    (dict(nref=3, n_sub=12, oversampling=1), "144 boundary rows: one-pass QR"),
    (dict(nref=4, n_sub=4, oversampling=2), "5x5-cell patches, 80 boundary rows: single chunk"),
    (dict(nref=4, n_sub=4, oversampling=3, dist="D100"), "49 coarse dofs: generic LDS QR / four-wave Jacobi"),
])
def test_selection_stage_paths(so, kw, what):
    """Every variant of the selection stage (register one-pass QR, chunked TSQR, generic LDS path,
    pivoted second stage + Jacobi on rim patches) against the oracle, on a sample of patches that
    covers every patch shape."""
    kw = dict(kw)
    dist = kw.pop("dist", "D1e4")
    cfg, g = _mk(so, stabilize=1, **kw)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    shapes = {}
    for pid in range(g.num_patches):
        info = g.patch_layout(pid)
        shapes.setdefault((info.mx, info.my, tuple(info.side_domain)), []).append(pid)
    ids = np.array(sorted(p for v in shapes.values() for p in v[:2]), dtype=np.uint32)
    basis, premult, offs = g.compute_basis(ids)
    if kw["oversampling"] < 3:
        for k, pid in enumerate(ids):
            _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), what)
        return
    # oversampling 3: rim patches have a continuum of singular values of G = BD'^T BD' running
    # through the reference's 1e-15 cutoff (LOD.cc:667), up to 16 of 48 are cut.  What is asserted:
    #  (1) the GPU takes exactly the oracle's decisions (n_cut, n_dropped; slod_plan_diagnostics);
    #  (2) |dphi| <= 1e-10 wherever the oracle itself is stable to 1e-10 under the rounding noise
    #      of another fp64 solver (so.selection_conditioning: 1e-13 relative noise on X, the level
    #      at which two direct solvers differ); elsewhere the patch is ill-conditioned for EVERY
    #      implementation (the reference's KLU + dgesdd included) and the bar is 10x that spread.
    plan = g.plan(ids, offs)
    import torch
    dev = torch.device("cuda", 0)
    b = torch.zeros(basis.size, dtype=torch.float64, device=dev)
    q = torch.zeros_like(b)
    plan.execute(b.data_ptr(), q.data_ptr())
    plan.status()
    dg = plan.diagnostics()
    assert np.array_equal(b.cpu().numpy(), basis)      # plan path == host-buffer path, bit for bit
    unstable = []
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid))
        phi0, _, diag = so.patch_basis(cfg, fields, int(pid))
        spread, stable = so.selection_conditioning(cfg, fields, int(pid))
        got = basis[int(offs[k]):int(offs[k]) + p.n_f]
        err = np.abs(got - phi0.ravel()).max()
        assert np.isfinite(got).all()
        if stable:
            assert (dg[k].n_cut, dg[k].n_dropped) == (diag.n_cut[0], diag.n_dropped[0]), \
                "%s patch %d: decisions gpu (%d,%d) oracle (%d,%d)" % (what, pid, dg[k].n_cut, dg[k].n_dropped,
                                                                       diag.n_cut[0], diag.n_dropped[0])
        tol = TOL_PHI if (stable and spread <= TOL_PHI) else max(TOL_PHI, 10.0 * spread)
        if tol > TOL_PHI:
            unstable.append((int(pid), spread, err))
        assert err <= tol, "%s patch %d: %.3e (tol %.1e, cut %d, oracle spread under solver noise %.3e)" % (
            what, pid, err, tol, diag.n_cut[0], spread)
    print("%s: %d of %d patches ill-conditioned for every fp64 implementation (pid, oracle spread, gpu err): %s"
          % (what, len(unstable), len(ids), ["%d %.1e %.1e" % u for u in unstable]))


@pytest.mark.parametrize("kw", [dict(nref=3, n_sub=4, oversampling=1), dict(nref=5, n_sub=8, oversampling=2)])
@pytest.mark.parametrize("dist", ["D100", "D1e4"])
def test_selection_decisions_match_oracle(so, kw, dist):
    """slod_plan_diagnostics on C1 and C2 (ALL patches): the discontinuous part of LOD.cc:656-725
    -- singular values under the 1e-15 cutoff, triplets put back by the 0.5-loop, final
    ||d||_inf -- equals the oracle's, patch by patch."""
    import torch
    cfg, g = _mk(so, stabilize=1, **kw)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    ids = np.arange(g.num_patches, dtype=np.uint32)
    plan = g.plan(ids)
    dev = torch.device("cuda", 0)
    b = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
    q = torch.zeros_like(b)
    plan.execute(b.data_ptr(), q.data_ptr())
    plan.status()
    dg = plan.diagnostics()
    sizes = np.array([so.patch_info(cfg, int(p)).n_f for p in ids], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    import ctypes as C
    diags = []
    for pid in ids:   # decisions of the oracle (full pipeline per patch, OpenMP not needed at this size)
        diags.append(so.patch_basis(cfg, fields, int(pid))[2])
    n_path2 = 0
    for k, pid in enumerate(ids):
        d0 = diags[k]
        assert dg[k].path in (1, 2)
        n_path2 += dg[k].path == 2
        assert (dg[k].n_cut, dg[k].n_dropped) == (d0.n_cut[0], d0.n_dropped[0]), "patch %d" % pid
        assert abs(dg[k].dinf - d0.dinf[0]) <= 1e-9, "patch %d" % pid
        if dg[k].path == 1:   # proven decision-free: the oracle agrees
            assert d0.n_cut[0] == 0 and d0.n_dropped[0] == 0
    print("%s %s: %d of %d patches replayed the truncation loop" % (kw, dist, n_path2, len(ids)))



def _decisions(g, ids, offs):
    """slod_plan_diagnostics of the same patches (plan path; outputs must equal the host-buffer path)"""
    import torch
    plan = g.plan(ids, offs)
    dev = torch.device("cuda", 0)
    b = torch.zeros(max(plan.output_size, 1), dtype=torch.float64, device=dev)
    q = torch.zeros_like(b)
    plan.execute(b.data_ptr(), q.data_ptr())
    torch.cuda.synchronize()
    plan.status()
    return plan.diagnostics()


def test_c3_patch_shapes(so):
    """C3's n_sub / oversampling (n_sub=16, l=3: up to 111 dofs per grid line, 49 coarse dofs) on
    an 8x8 coarse grid: one patch of EVERY shape of that grid.  No patch of an 8x8 grid is interior
    with l=3 -- its 7x7-cell patches touch two domain sides (225 id-99 boundary rows); C3's own
    classes (448 and 337 rows) are covered by tests/test_gpu_c3.py on a 16x16 grid.  Decisions
    (n_cut, n_dropped) must equal the oracle's wherever the oracle's are stable."""
    cfg, g = _mk(so, nref=3, n_sub=16, oversampling=3, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    shapes = {}
    for pid in range(g.num_patches):
        info = g.patch_layout(pid)
        shapes.setdefault((info.mx, info.my, tuple(info.side_domain)), []).append(pid)
    ids = np.array(sorted(v[0] for v in shapes.values()), dtype=np.uint32)
    assert any(g.patch_layout(int(p)).mx == 7 and g.patch_layout(int(p)).my == 7 for p in ids)
    basis, premult, offs = g.compute_basis(ids)
    dg = _decisions(g, ids, offs)
    worst, widened = 0.0, 0
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid))
        spread, stable = so.selection_conditioning(cfg, fields, int(pid))
        phi0, _, diag = so.patch_basis(cfg, fields, int(pid))
        err = np.abs(basis[int(offs[k]):int(offs[k]) + p.n_f] - phi0.ravel()).max()
        if stable:
            assert (dg[k].n_cut, dg[k].n_dropped) == (diag.n_cut[0], diag.n_dropped[0]), "patch %d" % pid
        tol = TOL_PHI if (stable and spread <= TOL_PHI) else max(TOL_PHI, 10.0 * spread)
        widened += tol > TOL_PHI
        assert err <= tol, "C3 shape patch %d (%dx%d): %.3e (tol %.1e)" % (pid, p.mx, p.my, err, tol)
        worst = max(worst, err)
    # builder log gpurun_out/r2_l3b.log: the widened class is a minority of rim patches
    assert widened <= len(ids) // 2, "%d of %d patches needed the widened tolerance" % (widened, len(ids))
    print("C3 shapes: %d patches, worst |dphi| %.3e, %d with widened tolerance" % (len(ids), worst, widened))


@pytest.mark.parametrize("n_sub", [8, 10, 12])
def test_large_line_blocks(so, n_sub):
    """Line blocks of 55 / 69 / 83 dofs (4, 5, 6 MFMA tiles; register tiles 8, 10, 12 of the
    VALU kernels) -- sizes between C2 and C3 that no BASELINE config hits."""
    cfg, g = _mk(so, nref=3, n_sub=n_sub, oversampling=3, stabilize=1)
    fields = make_fields(so, cfg, "D100")
    _upload(g, fields)
    ids = np.array([0, 3, 9, 27, 36], dtype=np.uint32)
    basis, premult, offs = g.compute_basis(ids)
    dg = _decisions(g, ids, offs)
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid))
        spread, stable = so.selection_conditioning(cfg, fields, int(pid))
        phi0, _, diag = so.patch_basis(cfg, fields, int(pid))
        err = np.abs(basis[int(offs[k]):int(offs[k]) + p.n_f] - phi0.ravel()).max()
        if stable:
            assert (dg[k].n_cut, dg[k].n_dropped) == (diag.n_cut[0], diag.n_dropped[0]), "patch %d" % pid
        tol = TOL_PHI if (stable and spread <= TOL_PHI) else max(TOL_PHI, 10.0 * spread)
        assert err <= tol, "n_sub %d patch %d: %.3e (tol %.1e)" % (n_sub, pid, err, tol)


@pytest.mark.gpu
def test_fused_and_split_selection_agree(so, monkeypatch):
    """SLOD_FUSE_SELECT=0 runs the selection stage as its own launch (k_select); the fused default
    runs the same device function at the end of k_solve_tw: bit-identical outputs."""
    cfg, g = _mk(so, nref=4, n_sub=4, oversampling=2, stabilize=1)
    fields = make_fields(so, cfg, "D1e4")
    _upload(g, fields)
    ids = np.arange(0, g.num_patches, 3, dtype=np.uint32)
    b1, p1, _ = g.compute_basis(ids)
    monkeypatch.setenv("SLOD_FUSE_SELECT", "0")
    b0, p0, _ = g.compute_basis(ids)
    assert np.array_equal(b0, b1) and np.array_equal(p0, p1)
    # same for the stencil assembly: SLOD_FUSE_ASSEMBLE=0 launches k_assemble on its own
    monkeypatch.setenv("SLOD_FUSE_ASSEMBLE", "0")
    b2, p2, _ = g.compute_basis(ids)
    assert np.array_equal(b2, b1) and np.array_equal(p2, p1)


@pytest.mark.gpu
def test_balanced_launch_order_is_transparent(so, monkeypatch):
    """The plan launches the patches in a cost-balanced order (SLOD_BALANCE=0: the caller's order);
    outputs, offsets and the per-patch diagnostics stay in the caller's order either way."""
    cfg, g = _mk(so, nref=4, n_sub=4, oversampling=2, stabilize=1)
    fields = make_fields(so, cfg, "D1e4")
    _upload(g, fields)
    ids = np.arange(g.num_patches - 1, -1, -1, dtype=np.uint32)[::2].copy()   # descending, every other patch
    b1, p1, _ = g.compute_basis(ids)
    plan = g.plan(ids)
    import torch
    dev = torch.device("cuda", 0)
    tb = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(tb)
    plan.execute(tb.data_ptr(), tq.data_ptr())
    plan.status()
    d1 = [(d.path, d.n_cut, d.n_dropped) for d in plan.diagnostics()]
    monkeypatch.setenv("SLOD_BALANCE", "0")
    b0, p0, _ = g.compute_basis(ids)
    plan0 = g.plan(ids)
    plan0.execute(tb.data_ptr(), tq.data_ptr())
    plan0.status()
    d0 = [(d.path, d.n_cut, d.n_dropped) for d in plan0.diagnostics()]
    assert np.array_equal(b0, b1) and np.array_equal(p0, p1)
    assert d0 == d1 and len(set(d1)) > 1          # rim and full patches decide differently


@pytest.mark.parametrize("kw,ids", [
    (dict(nref=3, n_sub=4, oversampling=1), None),
    (dict(nref=4, n_sub=4, oversampling=2), None),
    (dict(nref=4, n_sub=4, oversampling=3, dist="D100"), range(0, 256, 5)),
    (dict(nref=5, n_sub=8, oversampling=2), range(0, 1024, 13)),
])
def test_nested_dissection_solver(so, kw, ids, monkeypatch, capfd):
    """SLOD_SOLVE=nd: static condensation per cell + edge sets + skeleton lines (k_solve_nd) instead
    of the line-by-line elimination.  X = A_II^-1 P^T_I itself (slod_patch_solution) and the basis
    against the oracle, on every patch shape of the plan; the launch log proves the kernel ran."""
    monkeypatch.setenv("SLOD_SOLVE", "nd")
    monkeypatch.setenv("SLOD_DEBUG", "1")
    kw = dict(kw)
    dist = kw.pop("dist", "D1e4")
    cfg, g = _mk(so, stabilize=1, **kw)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    shapes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), pid)
    for pid in sorted(shapes.values()):
        X = g.patch_solution(pid)
        ref = so.patch_debug(cfg, fields, pid)["X"]
        assert np.abs(X - ref).max() <= 1e-11 * np.abs(ref).max(), "patch %d" % pid
    pids = np.array(sorted(set(shapes.values()) | set(ids if ids is not None else range(g.num_patches))), dtype=np.uint32)
    basis, premult, offs = g.compute_basis(pids)
    assert "k_solve_nd<" in capfd.readouterr().err
    for k, pid in enumerate(pids):
        if kw["oversampling"] >= 3:   # ill-conditioned rim patches: the rule of test_selection_stage_paths
            p = so.patch_info(cfg, int(pid))
            spread, stable = so.selection_conditioning(cfg, fields, int(pid))
            phi0, _, _ = so.patch_basis(cfg, fields, int(pid))
            err = np.abs(basis[int(offs[k]):int(offs[k]) + p.n_f] - phi0.ravel()).max()
            assert err <= (TOL_PHI if (stable and spread <= TOL_PHI) else max(TOL_PHI, 10.0 * spread)), "patch %d" % pid
        else:
            _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "nd")


@pytest.mark.parametrize("balance", ["1", "0"])
def test_chunked_plan_equals_single_launch(so, monkeypatch, balance):
    """SLOD_WORKSPACE_MB small enough that the plan runs in many workspace chunks (several launches,
    slots re-used, balanced launch order per plan): bit-identical to the one-launch plan, decisions
    and status included."""
    import torch
    monkeypatch.setenv("SLOD_BALANCE", balance)
    cfg, g = _mk(so, nref=4, n_sub=4, oversampling=2, stabilize=1)
    fields = make_fields(so, cfg, "D1e4")
    _upload(g, fields)
    ids = np.arange(g.num_patches, dtype=np.uint32)[::-1].copy()
    b1, p1, offs = g.compute_basis(ids)
    monkeypatch.setenv("SLOD_WORKSPACE_MB", "2")
    b2, p2, _ = g.compute_basis(ids)
    assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
    plan = g.plan(ids)
    dev = torch.device("cuda", 0)
    tb = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(tb)
    plan.execute(tb.data_ptr(), tq.data_ptr())
    plan.status()
    ms = plan.kernel_ms()
    assert ms[1] > 0.0
    dg = plan.diagnostics()
    for k in range(0, len(ids), 17):
        p = so.patch_info(cfg, int(ids[k]))
        assert np.array_equal(tb[k * plan.stride:k * plan.stride + p.n_f].cpu().numpy(), b1[int(offs[k]):int(offs[k]) + p.n_f])
        d0 = so.patch_basis(cfg, fields, int(ids[k]))[2]
        assert (dg[k].n_cut, dg[k].n_dropped) == (d0.n_cut[0], d0.n_dropped[0])


@pytest.mark.parametrize("dist", ["D100", "D1e4"])
def test_elasticity_c4_sample_with_projection_quirk(so, dist):
    """C4 as a dealii-slod elasticity run would configure it: adapter/LOD_hip.cc passes
    projection_quirk = (spacedim == 2), i.e. the row-parity component assignment of
    projection_P1_P0<2,2> (LODtools.h:43-67).  One patch of every shape plus every 16th patch."""
    cfg, g = _mk(so, nref=5, n_sub=8, oversampling=2, spacedim=2, stabilize=1, proj_quirk=1)
    fields = make_fields(so, cfg, dist)
    _upload(g, fields)
    shapes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), pid)
    ids = np.array(sorted(set(shapes.values()) | set(range(0, g.num_patches, 16))), dtype=np.uint32)
    basis, premult, offs = g.compute_basis(ids)
    worst = 0.0
    for k, pid in enumerate(ids):
        worst = max(worst, _check_patch(so, cfg, fields, int(pid), basis, premult, int(offs[k]), "C4 quirk")[0])
    print("C4 with projection quirk, %s: %d patches, worst |dphi| %.3e" % (dist, len(ids), worst))


def test_overlapped_executes_inside_the_library(so):
    """slod_plan_set_overlap(plan, 2): consecutive slod_plan_execute calls run on two internal streams with
    two workspaces.  Outputs bit-identical to the serial plan whatever buffer each execute writes;
    slod_plan_join orders a stream after them; status and decisions as in the serial run; plans in
    several workspace chunks refuse the mode."""
    import slod_amd
    import torch
    cfg, g = _mk(so, nref=4, n_sub=4, oversampling=2, stabilize=1)
    fields = make_fields(so, cfg, "D1e4")
    _upload(g, fields)
    ids = np.arange(g.num_patches, dtype=np.uint32)
    plan = g.plan(ids)
    dev = torch.device("cuda", 0)
    n = len(ids) * plan.stride
    ref_b = torch.zeros(n, dtype=torch.float64, device=dev)
    ref_q = torch.zeros_like(ref_b)
    plan.execute(ref_b.data_ptr(), ref_q.data_ptr())
    plan.status()
    d_ref = [(d.path, d.n_cut, d.n_dropped) for d in plan.diagnostics()]
    plan.set_overlap(2)
    outs = [(torch.zeros_like(ref_b), torch.zeros_like(ref_b)) for _ in range(3)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(7):
            b, q = outs[k % 3]
            b.zero_()                              # stream work the execute must be ordered after
            plan.execute(b.data_ptr(), q.data_ptr(), s.cuda_stream)
        plan.join(s.cuda_stream)
        sums = [float(b.sum()) for b, _ in outs]   # ordered after the executes by the join
    torch.cuda.synchronize()
    plan.status()
    for b, q in outs:
        assert torch.equal(b, ref_b) and torch.equal(q, ref_q)
    assert all(abs(x - float(ref_b.sum())) < 1e-9 for x in sums)
    assert [(d.path, d.n_cut, d.n_dropped) for d in plan.diagnostics()] == d_ref
    assert plan.kernel_ms()[1] > 0.0
    plan.set_overlap(1)
    b, q = outs[0]
    b.zero_()
    plan.execute(b.data_ptr(), q.data_ptr())
    plan.status()
    assert torch.equal(b, ref_b)


def test_overlap_refused_for_chunked_plans(so, monkeypatch):
    import slod_amd
    monkeypatch.setenv("SLOD_WORKSPACE_MB", "2")
    cfg, g = _mk(so, nref=4, n_sub=4, oversampling=2, stabilize=1)
    plan = g.plan(np.arange(g.num_patches, dtype=np.uint32))
    with pytest.raises(slod_amd.SlodError) as e:
        plan.set_overlap(2)
    assert e.value.code == -4
