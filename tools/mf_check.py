# Quick parity check of the patch-solve kernel families against the oracle (X = A^-1 P^T and phi).
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, slod_amd, slod_oracle as so

def sample_ids(g, per_shape=1):
    shapes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), []).append(pid)
    return np.array(sorted(p for v in shapes.values() for p in v[:per_shape]), dtype=np.uint32)

def run(kw, dist, maxp=12, solve_only=False):
    cfg = so.make_cfg(**kw); g = slod_amd.Slod(**kw)
    d, lo, hi = {"D100": (0, 1.0, 100.0), "D1e4": (1, 1.0, 1.0e4)}[dist]
    fields = [so.fill_coefficient(20250614 + f, d, lo, hi, g.NE) for f in range(kw.get("spacedim", 1))]
    for f, a in enumerate(fields):
        g.set_coefficient(f, a)
    allids = sample_ids(g)
    full = [p for p in allids if g.patch_layout(int(p)).mx == g.patch_layout(int(p)).my == 2 * kw['oversampling'] + 1][:3]
    ids = np.array(sorted(set(list(allids[:maxp - 3]) + full)), dtype=np.uint32)
    worst_x = worst_p = 0.0
    t0 = time.time()
    for pid in ids[:4]:
        X = g.patch_solution(int(pid)); ref = so.patch_debug(cfg, fields, int(pid))["X"]
        worst_x = max(worst_x, np.abs(X - ref).max() / np.abs(ref).max())
    basis, premult, offs = g.compute_basis(ids)
    s = cfg.spacedim
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid)); phi, psi, _ = so.patch_basis(cfg, fields, int(pid))
        got = basis[int(offs[k]):int(offs[k]) + s * p.n_f].reshape(s, p.n_f)
        worst_p = max(worst_p, np.abs(got - phi).max())
    print("%-4s %s %s: %d patches  rel|dX| %.2e  |dphi| %.2e  (%.1f s)" % (os.environ.get("SLOD_SOLVE", "auto"), kw, dist, len(ids), worst_x, worst_p, time.time() - t0), flush=True)

cases = [
    (dict(nref=2, n_sub=2, oversampling=1, spacedim=1, stabilize=1), "D100"),
    (dict(nref=3, n_sub=4, oversampling=1, spacedim=1, stabilize=1), "D100"),
    (dict(nref=3, n_sub=4, oversampling=1, spacedim=1, stabilize=0), "D1e4"),
    (dict(nref=4, n_sub=4, oversampling=2, spacedim=1, stabilize=1), "D1e4"),
    (dict(nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1), "D1e4"),
    (dict(nref=3, n_sub=3, oversampling=1, spacedim=1, stabilize=1), "D100"),
    (dict(nref=3, n_sub=5, oversampling=2, spacedim=1, stabilize=1), "D100"),
    (dict(nref=3, n_sub=8, oversampling=3, spacedim=1, stabilize=1), "D100"),
    (dict(nref=3, n_sub=4, oversampling=1, spacedim=2, stabilize=1), "D100"),
    (dict(nref=4, n_sub=4, oversampling=2, spacedim=2, stabilize=1), "D100"),
]
sel = sys.argv[1:]
for i, (kw, dist) in enumerate(cases):
    if sel and str(i) not in sel:
        continue
    run(kw, dist)
