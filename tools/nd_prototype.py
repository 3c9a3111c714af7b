"""numpy model of k_solve_nd (csrc/slod_solve_nd.hip): static condensation per virtual cell,
vertical-edge sets per cell row, then the horizontal skeleton lines as a dense block-tridiagonal
system.  Same index conventions as the kernel (line coordinates (l, i), cell ring numbering,
edge numbering), so every intermediate of the kernel can be compared with this model.
Checked against the oracle's X = A_II^{-1} P^T_I.  Development tool, not part of the product."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slod_oracle as so  # noqa: E402


class Patch:
    def __init__(self, cfg, fields, pid, nv=None):
        self.p = p = so.patch_info(cfg, pid)
        self.n = n = cfg.n_sub
        self.nv = nv or n
        self.st = so.assemble_patch(cfg, fields, pid)[:, :, 0, 0]   # [node][dir]
        self.PT = so.patch_pt(cfg, pid)
        self.nx, self.ny = p.nx, p.ny
        self.npx = p.nx + 1
        self.tr = p.nx > p.ny                      # lines run along y when x is the longer side
        self.m = (p.ny if self.tr else p.nx) - 1
        self.L = (p.nx if self.tr else p.ny) - 1
        self.Ca, self.Cb = (self.L + 1) // self.nv, (self.m + 1) // self.nv

    def node(self, l, i):
        ix, iy = (l + 1, i + 1) if self.tr else (i + 1, l + 1)
        return ix + iy * self.npx

    def coupling(self, l, i, dl, o):
        """A[(l,i),(l+dl,i+o)] (0 outside the interior)"""
        if not (0 <= i + o < self.m and 0 <= l + dl < self.L and 0 <= i < self.m and 0 <= l < self.L):
            return 0.0
        dx, dy = (dl, o) if self.tr else (o, dl)
        return self.st[self.node(l, i), (dy + 1) * 3 + dx + 1]

    def live(self, l, i):
        return 0 <= l < self.L and 0 <= i < self.m

    def ring(self, a, b):
        """boundary ring of cell (a,b): list of (l,i), kernel numbering"""
        nv = self.nv
        r = [(a * nv - 1, b * nv - 1 + j) for j in range(nv + 1)]
        r += [(a * nv + nv - 1, b * nv - 1 + j) for j in range(nv + 1)]
        r += [(a * nv + j, b * nv - 1) for j in range(nv - 1)]
        r += [(a * nv + j, b * nv + nv - 1) for j in range(nv - 1)]
        return r

    def interior(self, a, b):
        nv = self.nv
        return [(a * nv + li, b * nv + ii) for li in range(nv - 1) for ii in range(nv - 1)]

    def A(self, p, q):
        return self.coupling(p[0], p[1], q[0] - p[0], q[1] - p[1]) if max(abs(q[0] - p[0]), abs(q[1] - p[1])) <= 1 else 0.0

    def rhs(self, l, i):
        return self.PT[self.node(l, i), :]


def solve_nd(P):
    nv, m, L, Ca, Cb = P.nv, P.m, P.L, P.Ca, P.Cb
    nc = P.PT.shape[1]
    X = np.zeros((L, m, nc))
    # ---- level 0: cells
    cells = {}
    for a in range(Ca):
        for b in range(Cb):
            I, R = P.interior(a, b), P.ring(a, b)
            Acc = np.array([[P.A(p, q) for q in I] for p in I])
            Acs = np.array([[P.A(p, q) if P.live(*q) else 0.0 for q in R] for p in I])
            F = np.array([P.rhs(*p) for p in I])
            Y = np.linalg.solve(Acc, Acs)
            cells[a, b] = dict(C=-Acs.T @ Y, g=-Acs.T @ np.linalg.solve(Acc, F), Acc=Acc, Acs=Acs, F=F, I=I, R=R)
    # ---- skeleton numbering
    def hline(a):  # nodes of H line a
        return [(a * nv + nv - 1, i) for i in range(m)]

    def edges(a):
        return [(a * nv + li, b * nv + nv - 1) for b in range(Cb - 1) for li in range(nv - 1)]

    def assemble(rows, cols, strips):
        """direct stencil couplings + cell contributions of the cells in `strips`"""
        out = np.array([[P.A(p, q) for q in cols] for p in rows]) if len(rows) and len(cols) else np.zeros((len(rows), len(cols)))
        ri = {p: k for k, p in enumerate(rows)}
        ci = {q: k for k, q in enumerate(cols)}
        for a in strips:
            for b in range(Cb):
                c = cells[a, b]
                for j1, p in enumerate(c["R"]):
                    if p in ri:
                        for j2, q in enumerate(c["R"]):
                            if q in ci:
                                out[ri[p], ci[q]] += c["C"][j1, j2]
        return out

    def assemble_rhs(rows, strips):
        out = np.array([P.rhs(*p) for p in rows]) if len(rows) else np.zeros((0, nc))
        ri = {p: k for k, p in enumerate(rows)}
        for a in strips:
            for b in range(Cb):
                c = cells[a, b]
                for j1, p in enumerate(c["R"]):
                    if p in ri:
                        out[ri[p]] += c["g"][j1]
        return out

    nH = Ca - 1
    T = [assemble(hline(a), hline(a), [a, a + 1]) for a in range(nH)]
    B = [assemble(hline(a), hline(a + 1), [a + 1]) for a in range(nH - 1)]   # T_a x T_{a+1}
    G = [assemble_rhs(hline(a), [a, a + 1]) for a in range(nH)]
    # ---- level 1: edge sets of every strip
    E = {}
    for a in range(Ca):
        e = edges(a)
        hs = (hline(a - 1) if a > 0 else []) + (hline(a) if a < Ca - 1 else [])
        AEE = assemble(e, e, [a])
        AEH = assemble(e, hs, [a])
        GE = assemble_rhs(e, [a])
        if len(e):
            Y = np.linalg.solve(AEE, np.hstack([AEH, GE]))
        else:
            Y = np.zeros((0, len(hs) + nc))
        YH, YG = Y[:, :len(hs)], Y[:, len(hs):]
        U = AEH.T @ YH
        nb = m if a > 0 else 0
        if a > 0:
            T[a - 1] -= U[:nb, :nb]
            G[a - 1] -= AEH[:, :nb].T @ YG
        if a < Ca - 1:
            T[a] -= U[nb:, nb:]
            G[a] -= AEH[:, nb:].T @ YG
        if 0 < a < Ca - 1:
            B[a - 1] -= U[:nb, nb:]
        E[a] = dict(e=e, YH=YH, YG=YG, nb=nb)
    # ---- level 2: H lines, block tridiagonal
    V, W, Z = [], [], []
    S = None
    for a in range(nH):
        S = T[a] - (B[a - 1].T @ W[a - 1] if a > 0 else 0.0)
        V.append(np.linalg.inv(S))
        R = G[a] - (B[a - 1].T @ Z[a - 1] if a > 0 else 0.0)
        Z.append(V[a] @ R)
        W.append(V[a] @ B[a] if a < nH - 1 else None)
    XH = [None] * nH
    for a in range(nH - 1, -1, -1):
        XH[a] = Z[a] - (W[a] @ XH[a + 1] if a < nH - 1 else 0.0)
        for i in range(m):
            X[a * nv + nv - 1, i] = XH[a][i]
    # ---- back substitution: edges, then cells
    for a in range(Ca):
        d = E[a]
        xs = np.vstack(([XH[a - 1]] if a > 0 else []) + ([XH[a]] if a < Ca - 1 else [])) if nH > 0 else np.zeros((0, nc))
        xe = d["YG"] - d["YH"] @ xs
        for k, (l, i) in enumerate(d["e"]):
            X[l, i] = xe[k]
    for (a, b), c in cells.items():
        xs = np.array([X[l, i] if P.live(l, i) else np.zeros(nc) for (l, i) in c["R"]])
        xc = np.linalg.solve(c["Acc"], c["F"] - c["Acs"] @ xs)
        for k, (l, i) in enumerate(c["I"]):
            X[l, i] = xc[k]
    return X, dict(cells=cells, T=T, B=B, G=G, E=E, V=V)


if __name__ == "__main__":
    from conftest import make_fields
    for kw, pids in [(dict(nref=3, n_sub=4, oversampling=1), [0, 9, 27, 63]),
                     (dict(nref=4, n_sub=4, oversampling=2), [0, 5, 85, 100]),
                     (dict(nref=5, n_sub=8, oversampling=2), [0, 2, 341, 1023])]:
        cfg = so.make_cfg(stabilize=1, **kw)
        fields = make_fields(so, cfg, "D1e4")
        for pid in pids:
            P = Patch(cfg, fields, pid)
            X, _ = solve_nd(P)
            ref = so.patch_debug(cfg, fields, pid)["X"]
            Xf = np.zeros_like(ref)
            for l in range(P.L):
                for i in range(P.m):
                    Xf[P.node(l, i)] = X[l, i]
            print(kw, pid, (P.p.mx, P.p.my), "tr", P.tr, "err %.2e" % (np.abs(Xf - ref).max() / np.abs(ref).max()))
