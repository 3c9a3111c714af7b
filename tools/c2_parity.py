# Worst deviation of the HIP path from the oracle over ALL 1024 patches of the north-star config.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, slod_amd, slod_oracle as so
kw = dict(nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
cfg = so.make_cfg(**kw)
g = slod_amd.Slod(**kw)
for dist, args in (("D100", (0, 1.0, 100.0)), ("D1e4", (1, 1.0, 1.0e4))):
    field = so.fill_coefficient(20250614, args[0], args[1], args[2], g.NE)
    g.set_coefficient(0, field)
    ids = np.arange(g.num_patches, dtype=np.uint32)
    basis, premult, offs = g.compute_basis(ids)
    phi, psi = so.basis_many(cfg, [field], ids, offs.astype(np.int64), basis.size, nthreads=16)
    print(dist, "worst |dphi| %.3e   worst |dpsi| %.3e (|psi|max %.3e)" % (np.abs(basis - phi).max(), np.abs(premult - psi).max(), np.abs(psi).max()))
