"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/slod.h declares,
and its host-only index calculus matches the reference golden / the oracle. No compute calls."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_library_exports_every_declared_symbol():
    import slod_amd
    lib = slod_amd.load()
    names = slod_amd.declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert lib.slod_abi_version() == 5


def test_release_library_exports_exactly_the_header():
    """The release .so exports include/slod.h and nothing else: no timing-experiment hooks
    (slod_debug_read_ms lives only in lib/libslod_hip_diag.so, built with -DSLOD_ENABLE_DIAG)."""
    import subprocess
    import slod_amd
    out = subprocess.check_output(["nm", "-D", "--defined-only", slod_amd.LIB_PATH], text=True)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("slod_"))
    assert exported == slod_amd.declared_symbols()
    # and the phase-skip environment variable is not even referenced by the release binary
    blob = open(slod_amd.LIB_PATH, "rb").read()
    assert b"SLOD_DIAG" not in blob


def test_create_rejects_bad_config():
    import slod_amd
    with pytest.raises(slod_amd.SlodError) as e:
        slod_amd.Slod(nref=2, n_sub=2, oversampling=1, spacedim=3)
    assert e.value.code == -2
    with pytest.raises(slod_amd.SlodError):
        slod_amd.Slod(nref=2, n_sub=0, oversampling=1)


def test_patch_layout_matches_create_patch_01_golden():
    import slod_amd
    g = slod_amd.Slod(nref=5, n_sub=1, oversampling=4)
    lines = open(os.path.join(GOLDEN, "reference", "create_patch_01.output")).read().strip().split("\n")[1:]
    for ln in lines:
        pid = int(ln.split(":")[0][2:])
        cnt = int(ln.split("{")[1].split("}")[0])
        info = g.patch_layout(pid)
        assert info.mx * info.my == cnt
        assert len(g.patch_cells(pid)) == cnt


@pytest.mark.parametrize("kw", [dict(nref=2, n_sub=2, oversampling=1), dict(nref=3, n_sub=4, oversampling=2),
                                dict(nref=5, n_sub=8, oversampling=2), dict(nref=3, n_sub=3, oversampling=1, spacedim=2),
                                dict(n_cells=10, n_sub=7, oversampling=3)])
def test_patch_layout_matches_oracle(so, kw):
    import slod_amd
    g = slod_amd.Slod(**kw)
    cfg = so.make_cfg(**kw)
    assert g.num_patches == so.num_patches(cfg)
    for pid in range(0, g.num_patches, max(1, g.num_patches // 97)):
        a, b = g.patch_layout(pid), so.patch_info(cfg, pid)
        assert (a.cx, a.cy, a.x0, a.y0, a.mx, a.my, a.nx, a.ny) == (b.cx, b.cy, b.x0, b.y0, b.mx, b.my, b.nx, b.ny)
        assert list(a.side_domain) == list(b.side_domain)
        assert (a.n_fine, a.n_internal, a.n_boundary, a.n_coarse, a.is_lod) == (b.n_f, b.n_i, b.n_b, b.n_c, b.is_lod)
        assert g.patch_cells(pid) == so.patch_cells(cfg, pid)


def test_dof_permutation_is_bijection_and_first_cell_hierarchic():
    import slod_amd
    for s in (1, 2):
        g = slod_amd.Slod(nref=2, n_sub=3, oversampling=1, spacedim=s)
        for pid in (0, 5, 15):
            info = g.patch_layout(pid)
            perm = g.patch_dof_permutation(pid)
            assert sorted(perm.tolist()) == list(range(info.n_fine))
            # the first s*(n+1)^2 deal.II dofs are those of the centre cell, vertices first
            npx = info.nx + 1
            bx, by = (info.cx - info.x0) * 3, (info.cy - info.y0) * 3
            assert perm[0] == s * (bx + by * npx)
            assert perm[s] == s * (bx + 3 + by * npx)
            first = set(perm[:s * 16].tolist())
            want = {s * ((bx + i) + (by + j) * npx) + c for i in range(4) for j in range(4) for c in range(s)}
            assert first == want


def test_partition_is_evenly_distributed():
    import slod_amd
    for total, ranks in ((1024, 8), (1024, 3), (7, 4), (0, 2), (16384, 8)):
        spans = [slod_amd.partition(total, ranks, r) for r in range(ranks)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for a, b in zip(spans[:-1], spans[1:]):
            assert a[1] == b[0]
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_compute_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device the compute entry points return
    SLOD_ERR_DEVICE (on the GPU box this test is a no-op)."""
    import slod_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = slod_amd.Slod(nref=2, n_sub=2, oversampling=1)
    with pytest.raises(slod_amd.SlodError) as e:
        g.set_coefficient(0, np.ones(8 * 8 * 4))
    assert e.value.code == -3
