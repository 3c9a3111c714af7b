#!/usr/bin/env python3
"""Generates tests/golden/slod_fixtures.npz with the INDEPENDENT numpy/scipy restatement
(oracle/slod_numpy.py: dense PT / S_boundary, sparse LU, Gram matrix + LAPACK dgesdd, i.e. the
reference's literal formulation LOD.cc:345-767).  Run in the build container only:

    python tests/golden/make_fixtures.py

Each case stores its inputs (config, patch id, the patch's coefficient tile per field) and the
expected (phi, psi), so the GPU box needs neither this script's dependencies nor the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import slod_numpy as sn      # noqa: E402
import slod_oracle as so     # noqa: E402  (only its deterministic coefficient generator)

SEED = 20250614
CASES = []


def add(name, cfg, dist, pids):
    CASES.append((name, cfg, dist, pids))


small = dict(nref=2, n_sub=2, oversampling=1, spacedim=1)
add("small_lod_const", dict(small, stabilize=0), "const", list(range(16)))
add("small_lod_D100", dict(small, stabilize=0), "D100", list(range(16)))
add("small_slod_D100", dict(small, stabilize=1), "D100", list(range(16)))
add("small_lod_reuse", dict(small, stabilize=0, reuse_full=1), "D100", [5, 6, 9, 10])
c1 = dict(nref=3, n_sub=4, oversampling=1, spacedim=1, stabilize=1)
add("c1_slod_D100", c1, "D100", [0, 7, 9, 27, 36, 63])
c2 = dict(nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
# interior, edge (4x5), corner (3x3), near-edge, last
add("c2_slod_D100", c2, "D100", [51, 5, 0, 400, 1023])
add("c2_slod_D1e4", c2, "D1e4", [51, 5, 400])
el = dict(nref=2, n_sub=4, oversampling=1, spacedim=2, stabilize=1)
add("elast_slod_D100", el, "D100", [0, 5, 10, 15])
add("elast_lod_D100", dict(el, stabilize=0), "D100", [5, 15])
add("elast_slod_quirk", dict(el, proj_quirk=1), "D100", [5, 15])


def main():
    out = {}
    index = []
    for name, cfg, dist, pids in CASES:
        ocfg = so.make_cfg(**cfg)
        N = so.n_cells_per_side(ocfg)
        n = cfg["n_sub"]
        NE = N * n
        s = cfg["spacedim"]
        if dist == "const":
            fields = [np.ones(NE * NE * 4) for _ in range(s)]
        else:
            d, lo, hi = {"D100": (0, 1.0, 100.0), "D1e4": (1, 1.0, 1.0e4)}[dist]
            fields = [so.fill_coefficient(SEED + f, d, lo, hi, NE) for f in range(s)]
        f3 = [f.reshape(NE, NE, 4) for f in fields]
        for pid in pids:
            geo = sn.patch_geometry(cfg, pid)
            phi, psi, dbg = sn.patch_basis(cfg, f3, pid, svd_mode="gram", return_debug=True)
            phi2, _ = sn.patch_basis(cfg, f3, pid, svd_mode="stable")
            # the literal (Gram/dgesdd) and the stable (SVD of BD') formulation must agree here,
            # otherwise the case is too ill-conditioned to serve as a golden vector
            dev = np.abs(phi - phi2).max()
            assert dev < 5e-10, (name, pid, dev)
            ox, oy = geo["x0"] * n, geo["y0"] * n
            if cfg.get("reuse_full", 0) and geo["mx"] == 3 and geo["my"] == 3:
                g0 = sn.patch_geometry(cfg, sn.first_full_patch(cfg))
                ox, oy = g0["x0"] * n, g0["y0"] * n
            nx, ny = n * geo["mx"], n * geo["my"]
            key = "%s/%d" % (name, pid)
            out[key + "/phi"] = phi
            out[key + "/psi"] = psi
            for f in range(s):
                out[key + "/tile%d" % f] = f3[f][oy:oy + ny, ox:ox + nx, :].copy()
            out[key + "/tile_origin"] = np.array([ox, oy])
            out[key + "/n_dropped"] = np.array(dbg["n_dropped"] if dbg["n_dropped"] else [0] * s)
            index.append((key, cfg, pid, dev))
            print("%-28s dropped %s  |gram-stable| %.1e" % (key, dbg["n_dropped"], dev))
    out["index"] = np.array([k for k, _, _, _ in index])
    out["configs"] = np.array([repr(sorted(c.items())) for _, c, _, _ in index])
    out["pids"] = np.array([p for _, _, p, _ in index])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "slod_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
