"""Independent numpy/scipy restatement of the reference's per-patch algorithm.
TEST INFRASTRUCTURE ONLY (validates oracle/slod_oracle.c and generates tests/golden/*.npz).

It follows /root/reference/source/LOD.cc:345-767 LITERALLY: dense PT / PT_boundary /
S_boundary, index lists from fill_dofs_indices_vector, clear_row()ed sparse matrix solved
with a sparse LU (scipy SuperLU in place of Amesos-KLU), Gram matrix G = BD'^T BD' fed
to numpy.linalg.svd (LAPACK dgesdd -- the routine behind LAPACKFullMatrix::compute_svd).
`svd_mode="stable"` instead takes the SVD of BD' itself (what the C oracle and the HIP
kernels do); both are mathematically the same least-squares problem.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

G0 = 0.5 * (1.0 - 1.0 / np.sqrt(3.0))
G1 = 0.5 * (1.0 + 1.0 / np.sqrt(3.0))


def n_cells(cfg):
    return cfg["n_cells"] if cfg.get("n_cells", 0) > 0 else 2 ** cfg["nref"]


def morton_to_cell(nref, pid):
    x = y = 0
    for b in range(nref):
        x |= ((pid >> (2 * b)) & 1) << b
        y |= ((pid >> (2 * b + 1)) & 1) << b
    return x, y


def patch_geometry(cfg, pid):
    """LOD.cc:140-181 (cells, centre first) and LOD.cc:830-843 (side boundary ids)."""
    N, l = n_cells(cfg), cfg["oversampling"]
    if cfg.get("n_cells", 0) > 0:
        cx, cy = pid % N, pid // N
    else:
        cx, cy = morton_to_cell(cfg["nref"], pid)
    cells = [(cx, cy)]
    for lr in range(-l, l + 1):
        if 0 <= cx + lr < N:
            for lc in range(-l, l + 1):
                if 0 <= cy + lc < N and not (lr == 0 and lc == 0):
                    cells.append((cx + lr, cy + lc))
    xs = [c[0] for c in cells]
    ys = [c[1] for c in cells]
    x0, x1, y0, y1 = min(xs), max(xs), min(ys), max(ys)
    side_domain = (x0 == 0, x1 == N - 1, y0 == 0, y1 == N - 1)
    return dict(cx=cx, cy=cy, cells=cells, x0=x0, y0=y0, mx=x1 - x0 + 1, my=y1 - y0 + 1,
                side_domain=side_domain)


def shape_grads(xi, eta):
    gx = np.array([-(1 - eta), (1 - eta), -eta, eta])
    gy = np.array([-(1 - xi), -xi, (1 - xi), xi])
    return gx, gy


def element_matrix(s, coefs_q):
    """coefs_q: list of arrays [4] (alpha) or (lambda, mu). Diffusion.h:181-186 / Elasticity.h:246-258"""
    K = np.zeros((4 * s, 4 * s))
    for q in range(4):
        xi = G1 if (q & 1) else G0
        eta = G1 if (q & 2) else G0
        gx, gy = shape_grads(xi, eta)
        g = np.stack([gx, gy])  # [comp][node]
        jxw = 0.25
        if s == 1:
            K += coefs_q[0][q] * (np.outer(gx, gx) + np.outer(gy, gy)) * jxw
        else:
            lam, mu = coefs_q[0][q], coefs_q[1][q]
            for i in range(4):
                for a in range(2):
                    eps_i = np.zeros((2, 2))
                    eps_i[a, :] += 0.5 * g[:, i]
                    eps_i[:, a] += 0.5 * g[:, i]
                    div_i = g[a, i]
                    for j in range(4):
                        for b in range(2):
                            eps_j = np.zeros((2, 2))
                            eps_j[b, :] += 0.5 * g[:, j]
                            eps_j[:, b] += 0.5 * g[:, j]
                            div_j = g[b, j]
                            K[2 * i + a, 2 * j + b] += (2 * mu * np.sum(eps_i * eps_j)
                                                        + lam * div_i * div_j) * jxw
    return K


def first_full_patch(cfg):
    full = 2 * cfg["oversampling"] + 1
    for pid in range(n_cells(cfg) ** 2):
        g = patch_geometry(cfg, pid)
        if g["mx"] == full and g["my"] == full:
            return pid
    return None


def patch_basis(cfg, coefs, pid, svd_mode="gram", return_debug=False):
    """coefs: list of global per-qp fields, each [NE, NE, 4] (ey, ex, q). Returns phi[s,n_f], psi[s,n_f]."""
    N, n, s, l = n_cells(cfg), cfg["n_sub"], cfg.get("spacedim", 1), cfg["oversampling"]
    H = 1.0 / N
    h = H / n
    geo = patch_geometry(cfg, pid)
    mx, my, x0, y0 = geo["mx"], geo["my"], geo["x0"], geo["y0"]
    nx, ny = n * mx, n * my
    npx = nx + 1
    n_nodes = npx * (ny + 1)
    nf = s * n_nodes
    nc = s * mx * my
    ox, oy = x0 * n, y0 * n
    if cfg.get("reuse_full", 0) and mx == 2 * l + 1 and my == 2 * l + 1:
        g0 = patch_geometry(cfg, first_full_patch(cfg))
        ox, oy = g0["x0"] * n, g0["y0"] * n

    # --- unconstrained stiffness (LOD.cc:440-444)
    rows, cols, vals = [], [], []
    for ey in range(ny):
        for ex in range(nx):
            cq = [c[oy + ey, ox + ex, :] for c in coefs]
            K = element_matrix(s, cq)
            nodes = [(ex + (a & 1)) + (ey + (a >> 1)) * npx for a in range(4)]
            dofs = [s * nd + c for nd in nodes for c in range(s)]
            for i, di in enumerate(dofs):
                for j, dj in enumerate(dofs):
                    rows.append(di)
                    cols.append(dj)
                    vals.append(K[i, j])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(nf, nf)).tocsr()

    # --- dof index sets (LODtools.h:334-375)
    sd = geo["side_domain"]
    on0 = np.zeros(n_nodes, bool)
    on99 = np.zeros(n_nodes, bool)
    for iy in range(ny + 1):
        for ix in range(nx + 1):
            k = ix + iy * npx
            for cond, dom in ((ix == 0, sd[0]), (ix == nx, sd[1]), (iy == 0, sd[2]), (iy == ny, sd[3])):
                if cond:
                    if dom:
                        on0[k] = True
                    else:
                        on99[k] = True
    dof_on0 = np.repeat(on0, s)
    dof_on99 = np.repeat(on99, s)
    boundary = np.nonzero(dof_on99)[0]
    domain_boundary = np.nonzero(dof_on0)[0]
    internal = np.nonzero(~(dof_on0 | dof_on99))[0]

    # --- PT (LODtools.h:7-73, LOD.cc:341,478-495)
    PT = np.zeros((nf, nc))
    for k, (ccx, ccy) in enumerate(geo["cells"]):
        kx, ky = ccx - x0, ccy - y0
        if s == 2 and cfg.get("proj_quirk", 0):
            r = 0
            loc = []
            for v in range(4):
                for c in range(2):
                    loc.append(((v & 1) * n, (v >> 1) * n, c, 1.0))
            for L in range(4):
                for c in range(2):
                    for t in range(n - 1):
                        ij = [(0, t + 1), (n, t + 1), (t + 1, 0), (t + 1, n)][L]
                        loc.append((ij[0], ij[1], c, 2.0))
            for c in range(2):
                for t in range((n - 1) ** 2):
                    loc.append((1 + t % (n - 1), 1 + t // (n - 1), c, 4.0))
            for r, (jx, jy, c, w) in enumerate(loc):
                node = (kx * n + jx) + (ky * n + jy) * npx
                PT[2 * node + c, 2 * k + (r % 2)] += w * h * h / 4
        else:
            for jy in range(n + 1):
                for jx in range(n + 1):
                    w = (1.0 if jx in (0, n) else 2.0) * (1.0 if jy in (0, n) else 2.0)
                    node = (kx * n + jx) + (ky * n + jy) * npx
                    for c in range(s):
                        PT[s * node + c, s * k + c] += w * h * h / 4
    PT_boundary = PT[boundary, :].copy()          # LOD.cc:502-504
    PT[boundary, :] = 0.0                         # LOD.cc:512-518
    PT[domain_boundary, :] = 0.0
    S_boundary = A[boundary, :][:, internal].toarray()   # LOD.cc:524-526

    # --- clear_row (LOD.cc:537-543)
    def clear_rows(Acsr, rws):
        Al = Acsr.tolil()
        for j in rws:
            Al.rows[j] = [j]
            Al.data[j] = [1.0]
        return Al.tocsr()

    A_semi = clear_rows(A, domain_boundary)
    A_c = clear_rows(A_semi, boundary)
    Ainv_PT = spla.splu(A_c.tocsc()).solve(PT)     # LOD.cc:546
    P_Ainv_PT = PT.T @ Ainv_PT / H ** 2           # LOD.cc:548-551
    M = P_Ainv_PT.copy()
    D = np.linalg.inv(P_Ainv_PT)                  # LOD.cc:553

    is_lod = (not cfg["stabilize"]) or l == 0 or (mx * my == N * N)
    phi = np.zeros((s, nf))
    psi = np.zeros((s, nf))
    dbg = dict(M=M, D=D, X=Ainv_PT, bdofs=boundary, n_dropped=[], sigma=[])
    if not is_lod:
        Xi = Ainv_PT[internal, :]
        B_full = S_boundary @ Xi                   # LOD.cc:612
        BD = B_full @ D + (-PT_boundary) @ D       # LOD.cc:616-618
        dbg["BD"] = BD
    for d in range(s):
        if is_lod:
            sel = Ainv_PT @ D[:, d]                # LOD.cc:576-588
        else:
            b0 = BD[:, d]
            other = [j for j in range(nc) if j != d]
            newBD = BD[:, other]
            if svd_mode == "gram":
                G = newBD.T @ newBD                # LOD.cc:660
                g = newBD.T @ b0                   # LOD.cc:662
                U, sig, Vt = np.linalg.svd(G)      # LOD.cc:667 (dgesdd)
                inv = np.where(sig > 1e-15 * sig[0], 1.0 / np.where(sig > 0, sig, 1.0), 0.0)
                utg = U.T @ g
                terms = [Vt[i, :] * (utg[i] * inv[i]) for i in range(len(sig))]
            else:
                Ub, sb, Vt = np.linalg.svd(newBD, full_matrices=False)
                sig = sb ** 2
                keep = sig > 1e-15 * sig[0]
                coef = np.where(keep, (Ub.T @ b0) / np.where(sb > 0, sb, 1.0), 0.0)
                terms = [Vt[i, :] * coef[i] for i in range(len(sig))]
            d_i = -np.sum(terms, axis=0)
            dropped = 0
            for i in range(len(sig) - 1, -1, -1):  # LOD.cc:703-725
                if np.max(np.abs(d_i)) < 0.5:
                    break
                d_i = d_i + terms[i]
                dropped += 1
            dbg["n_dropped"].append(dropped)
            dbg["sigma"].append(sig)
            c_i = D[:, d].copy()                   # LOD.cc:727-743
            for idx, o in enumerate(other):
                c_i += d_i[idx] * D[:, o]
            sel = np.zeros(nf)
            sel[internal] = Xi @ c_i               # LOD.cc:745-750
        sel = sel / np.linalg.norm(sel)            # LOD.cc:752 / 591
        phi[d] = sel
        psi[d] = A_semi @ sel                      # LOD.cc:762-764
    if return_debug:
        return phi, psi, dbg
    return phi, psi


def solve_poisson_on_patch(n_rep, n_sub, cell, overlap):
    """tests/solve_poisson_problem_on_patch_01.cc: -lap u = 1 on the patch, zero Dirichlet on
    all four patch sides; returns the (n_rep*n_sub+1)^2 global lexicographic vector."""
    cfg = dict(n_cells=n_rep, nref=0, n_sub=n_sub, oversampling=overlap, spacedim=1, stabilize=0)
    pid = cell[0] + cell[1] * n_rep
    geo = patch_geometry(cfg, pid)
    nx, ny = n_sub * geo["mx"], n_sub * geo["my"]
    npx = nx + 1
    h = 1.0 / (n_rep * n_sub)
    K = element_matrix(1, [np.ones(4)])
    rows, cols, vals = [], [], []
    rhs = np.zeros(npx * (ny + 1))
    for ey in range(ny):
        for ex in range(nx):
            nodes = [(ex + (a & 1)) + (ey + (a >> 1)) * npx for a in range(4)]
            for i in range(4):
                rhs[nodes[i]] += h * h / 4
                for j in range(4):
                    rows.append(nodes[i]); cols.append(nodes[j]); vals.append(K[i, j])
    A = sp.coo_matrix((vals, (rows, cols))).tocsr()
    ix = np.arange(npx * (ny + 1)) % npx
    iy = np.arange(npx * (ny + 1)) // npx
    interior = np.nonzero((ix > 0) & (ix < nx) & (iy > 0) & (iy < ny))[0]
    u = np.zeros(npx * (ny + 1))
    u[interior] = spla.spsolve(A[interior, :][:, interior].tocsc(), rhs[interior])
    NG = n_rep * n_sub + 1
    out = np.zeros(NG * NG)
    gx = geo["x0"] * n_sub + ix
    gy = geo["y0"] * n_sub + iy
    out[gx + gy * NG] = u
    return out
