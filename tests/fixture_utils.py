"""Loads tests/golden/slod_fixtures.npz (made by tests/golden/make_fixtures.py from the
independent numpy/scipy restatement of the reference algorithm)."""
import ast
import os

import numpy as np

from conftest import GOLDEN

_FIX = None


def fixtures():
    global _FIX
    if _FIX is None:
        z = np.load(os.path.join(GOLDEN, "slod_fixtures.npz"))
        cases = []
        for key, cfg_s, pid in zip(z["index"], z["configs"], z["pids"]):
            key = str(key)
            cfg = dict(ast.literal_eval(str(cfg_s)))
            s = cfg["spacedim"]
            cases.append(dict(key=key, cfg=cfg, pid=int(pid), phi=z[key + "/phi"], psi=z[key + "/psi"],
                              tiles=[z[key + "/tile%d" % f] for f in range(s)],
                              origin=z[key + "/tile_origin"], n_dropped=z[key + "/n_dropped"]))
        _FIX = cases
    return _FIX


def global_fields(case):
    """A global per-qp field per component that equals the stored tile on the patch (the patch
    reads nothing else) and 1 elsewhere."""
    cfg = case["cfg"]
    N = cfg.get("n_cells", 0) or 2 ** cfg["nref"]
    NE = N * cfg["n_sub"]
    out = []
    ox, oy = [int(v) for v in case["origin"]]
    for t in case["tiles"]:
        f = np.ones((NE, NE, 4))
        f[oy:oy + t.shape[0], ox:ox + t.shape[1], :] = t
        out.append(f.ravel())
    return out


def tolerances(case):
    """phi: unit-norm vectors. The fixtures come from a sparse LU (SuperLU) + dgesdd pipeline;
    against a Cholesky-type solve the two differ by cond(A_II)*eps: <= 1e-12 at contrast <= 100,
    <= 5e-10 at contrast 1e4 (measured 5e-11).  psi = A_semi phi carries that times ||A||_inf."""
    hi = max(float(t.max()) for t in case["tiles"])
    return 5e-10 if hi > 1000.0 else 1e-12
