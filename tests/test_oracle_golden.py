"""The CPU oracle against the reference's OWN golden outputs (tests/golden/reference/*.output are
verbatim data files of /root/reference/tests) -- this is what pins the oracle."""
import os

import numpy as np

from conftest import GOLDEN

REF = os.path.join(GOLDEN, "reference")


def _parse_formatted(block, n, width=11):
    M = np.zeros((n, n))
    lines = [ln for ln in block.split("\n") if len(ln) > 0]
    assert len(lines) == n
    for i, ln in enumerate(lines):
        for j in range(n):
            f = ln[j * width:(j + 1) * width].strip()
            M[i, j] = float(f) if f else 0.0
    return M


def test_fe_q_iso_q1_01_cell_matrices(so):
    """tests/fe_q_iso_q1_01.cc: FE_Q_iso_Q1(3) Laplace cell matrix, hierarchic numbering, 3 decimals."""
    blocks = open(os.path.join(REF, "fe_q_iso_q1_01.output")).read().split("\n\n")
    g1 = _parse_formatted(blocks[0], 4)
    g2 = _parse_formatted(blocks[2], 16)
    assert np.array_equal(_parse_formatted(blocks[1], 4), g1)       # full loop == sub-element loop
    assert np.array_equal(_parse_formatted(blocks[3], 16), g2)
    assert np.abs(so.fe_q_iso_q1_cell_matrix(1, 3) - g1).max() <= 5e-4
    M2 = so.fe_q_iso_q1_cell_matrix(2, 3)
    assert np.abs(M2 - g2).max() <= 5e-4
    assert np.array_equal(M2 != 0, g2 != 0) or np.abs(M2[(M2 != 0) != (g2 != 0)]).max() < 1e-15


def test_create_patch_01_sizes_and_order(so):
    """tests/create_patch_01.cc: 32x32 grid, oversampling 4; patch i (Morton order) has {#cells}."""
    cfg = so.make_cfg(nref=5, n_sub=1, oversampling=4)
    lines = open(os.path.join(REF, "create_patch_01.output")).read().strip().split("\n")[1:]
    assert len(lines) == 1024
    for ln in lines:
        pid = int(ln.split(":")[0][2:])
        cnt = int(ln.split("{")[1].split("}")[0])
        p = so.patch_info(cfg, pid)
        cells = so.patch_cells(cfg, pid)
        assert p.mx * p.my == cnt and len(cells) == cnt
        assert cells[0] == p.cx + 32 * p.cy          # centre first (LOD.cc:151-154)
        assert len(set(cells)) == cnt


def test_solve_poisson_problem_on_patch_01(so):
    """tests/solve_poisson_problem_on_patch_01.cc: 10x10 grid, 7 subdivisions, patch around cell
    (1,4) with overlap 3, -lap u = 1, zero Dirichlet on the patch boundary; golden printed with
    4 significant digits (5041 values, 1632 non-zeros)."""
    gold = np.array([float(x) for x in
                     open(os.path.join(REF, "solve_poisson_problem_on_patch_01.output")).read().split()])
    assert gold.size == 71 * 71
    cfg = so.make_cfg(n_cells=10, n_sub=7, oversampling=3, stabilize=0)
    pid = 1 + 4 * 10
    p = so.patch_info(cfg, pid)
    assert (p.x0, p.y0, p.mx, p.my) == (0, 1, 5, 7)
    st = so.assemble_patch(cfg, [np.ones(70 * 70 * 4)], pid)
    h = 1.0 / 70
    u = so.solve_interior(p.nx, p.ny, 1, st, np.full((p.n_f, 1), h * h))[:, 0]
    out = np.zeros(71 * 71)
    ix = np.arange(p.n_f) % (p.nx + 1)
    iy = np.arange(p.n_f) // (p.nx + 1)
    out[(p.x0 * 7 + ix) + (p.y0 * 7 + iy) * 71] = u
    nz = gold != 0
    assert nz.sum() == 1632
    assert np.array_equal(nz, out != 0)
    assert np.max(np.abs(out[nz] - gold[nz]) / np.abs(gold[nz])) < 6e-4   # print rounding
    assert abs(out.max() - 2.417e-2) < 1e-5
    # the same solution as the sum of the columns of X = A^-1 P^T (sum_k P^T[:,k] = int phi_i)
    X = so.patch_debug(cfg, [np.ones(70 * 70 * 4)], pid)["X"]
    assert np.abs(X.sum(axis=1) - u).max() < 1e-15


def test_poisson_lod_example_load_vector(so):
    """tests/Poisson_LOD_Example.output: 16 patches of sizes (4,9) and 'fem rhs l2 norm =
    0.109375' (f = 1, H = 1/4, n = 2: the lumped load int phi_i on the 49 interior nodes)."""
    txt = open(os.path.join(REF, "Poisson_LOD_Example.output")).read()
    assert "number of patches = 16" in txt and "Patches size in (4, 9)" in txt
    cfg = so.make_cfg(nref=2, n_sub=2, oversampling=1, stabilize=0)
    sizes = [so.patch_info(cfg, p).mx * so.patch_info(cfg, p).my for p in range(16)]
    assert min(sizes) == 4 and max(sizes) == 9
    # the patch of cell (1,1) covers [0,3/4]^2; sum_k P^T[:,k] is the lumped load vector
    load = np.zeros((9, 9))
    for pid in range(16):
        p = so.patch_info(cfg, pid)
        PT = so.patch_pt(cfg, pid)
        col0 = PT[:, 0].reshape(p.ny + 1, p.nx + 1)     # column 0 = centre cell
        jx, jy = (p.cx - p.x0) * 2, (p.cy - p.y0) * 2
        load[p.cy * 2:p.cy * 2 + 3, p.cx * 2:p.cx * 2 + 3] += col0[jy:jy + 3, jx:jx + 3]
    assert abs(np.linalg.norm(load[1:-1, 1:-1]) - 0.109375) < 1e-15
