"""k_solve_nd against the oracle: X = A_II^-1 P^T_I (slod_patch_solution) and the full basis."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "dealii-slod_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("SLOD_SOLVE", "nd")
import numpy as np
import slod_amd
import slod_oracle as so
from conftest import make_fields

cases = [(dict(nref=3, n_sub=4, oversampling=1), [0, 9, 27, 63]),
         (dict(nref=4, n_sub=4, oversampling=2), [0, 5, 85, 100]),
         (dict(nref=4, n_sub=4, oversampling=3), [0, 5, 85, 100]),
         (dict(nref=5, n_sub=8, oversampling=2), [0, 2, 341, 1023, 500])]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for ci, (kw, pids) in enumerate(cases):
    if only >= 0 and ci != only:
        continue
    cfg = so.make_cfg(stabilize=1, **kw)
    g = slod_amd.Slod(stabilize=1, **kw)
    fields = make_fields(so, cfg, "D1e4")
    g.set_coefficient(0, fields[0])
    for pid in pids:
        X = g.patch_solution(pid)
        ref = so.patch_debug(cfg, fields, pid)["X"]
        err = np.abs(X - ref).max() / np.abs(ref).max()
        p = so.patch_info(cfg, pid)
        print(kw, pid, (p.mx, p.my), "X rel err %.2e" % err, flush=True)
        if not err < 1e-10:
            npx = p.nx + 1
            bad = np.argwhere(np.abs(X - ref) > 1e-8 * np.abs(ref).max())
            rows = sorted(set(int(b[0]) for b in bad))
            print("  bad rows (ix,iy):", [(r % npx, r // npx) for r in rows][:40], "n", len(rows))
    ids = np.arange(g.num_patches, dtype=np.uint32)
    t = time.time()
    b, q, offs = g.compute_basis(ids)
    worst = 0.0
    for k in ids[:: max(1, len(ids) // 40)]:
        phi, psi, _ = so.patch_basis(cfg, fields, int(k))
        n = phi.size
        worst = max(worst, np.abs(b[int(offs[k]):int(offs[k]) + n] - phi.ravel()).max())
    print(kw, "all patches: worst |dphi| %.2e (%.1f s)" % (worst, time.time() - t), flush=True)
