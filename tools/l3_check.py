# Diagnostic: oversampling 3 (49 coarse dofs, generic selection path): GPU vs oracle per patch shape.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, slod_amd, slod_oracle as so
kw = dict(nref=4, n_sub=4, oversampling=3, spacedim=1, stabilize=1)
cfg = so.make_cfg(**kw)
g = slod_amd.Slod(**kw)
field = so.fill_coefficient(20250614, 0, 1.0, 100.0, g.NE)
g.set_coefficient(0, field)
shapes = {}
for pid in range(g.num_patches):
    i = g.patch_layout(pid)
    shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), []).append(pid)
ids = np.array(sorted(p for v in shapes.values() for p in v[:2]), dtype=np.uint32)
basis, premult, offs = g.compute_basis(ids)
for k, pid in enumerate(ids):
    p = so.patch_info(cfg, int(pid))
    phi0, _, d = so.patch_basis(cfg, [field], int(pid))
    so.set_svd_mode(1); phi1, _, _ = so.patch_basis(cfg, [field], int(pid)); so.set_svd_mode(0)
    got = basis[int(offs[k]):int(offs[k]) + p.n_f]
    print(pid, (p.mx, p.my), list(p.side_domain), "n_c", p.n_c, "n_b", p.n_b, "cut", d.n_cut[0], "drop", d.n_dropped[0],
          "gpu-oracle %.2e" % np.abs(got - phi0.ravel()).max(), "gram-oracle %.2e" % np.abs(phi1 - phi0).max())
