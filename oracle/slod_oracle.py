"""ctypes binding of the C oracle (oracle/slod_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (dealii-slod_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslod_oracle.so")


class Cfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nref", "n_cells", "n_sub", "oversampling", "spacedim",
                                       "stabilize", "reuse_full", "proj_quirk")]


class Patch(C.Structure):
    _fields_ = [("pid", C.c_int), ("cx", C.c_int), ("cy", C.c_int), ("x0", C.c_int), ("y0", C.c_int),
                ("mx", C.c_int), ("my", C.c_int), ("nx", C.c_int), ("ny", C.c_int),
                ("side_domain", C.c_int * 4), ("n_f", C.c_int), ("n_i", C.c_int), ("n_b", C.c_int),
                ("n_c", C.c_int), ("is_lod", C.c_int)]


class Diag(C.Structure):
    _fields_ = [("n_dropped", C.c_int * 2), ("n_cut", C.c_int * 2), ("dinf", C.c_double * 2),
                ("sigma_max", C.c_double * 2), ("sigma_min", C.c_double * 2),
                ("cond_hint", C.c_double)]


def build(force=False):
    src = os.path.join(_HERE, "slod_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        _lib.so_patch_basis.argtypes = [C.POINTER(Cfg), C.POINTER(dp), C.c_int, dp, dp, C.POINTER(Diag)]
        _lib.so_patch_debug.argtypes = [C.POINTER(Cfg), C.POINTER(dp), C.c_int, dp, dp, dp,
                                        C.POINTER(C.c_int), dp]
        _lib.so_basis_many.argtypes = [C.POINTER(Cfg), C.POINTER(dp), C.POINTER(C.c_int), C.c_int, dp, dp,
                                       C.POINTER(C.c_longlong), C.c_int]
        _lib.so_assemble_patch.argtypes = [C.POINTER(Cfg), C.POINTER(Patch), C.POINTER(dp), dp]
        _lib.so_solve_interior.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp]
        _lib.so_patch_solve.argtypes = [C.POINTER(Cfg), C.POINTER(Patch), dp, dp]
        _lib.so_fill_coefficient.argtypes = [C.c_ulonglong, C.c_int, C.c_double, C.c_double, C.c_int, dp]
        _lib.so_fill_coefficient_rand.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, dp]
        _lib.so_local_matrix_poisson.argtypes = [dp, dp]
        _lib.so_local_matrix_elasticity.argtypes = [dp, dp, dp]
        _lib.so_fe_q_iso_q1_cell_matrix.argtypes = [C.c_int, C.c_int, dp]
        _lib.so_patch_init.argtypes = [C.POINTER(Cfg), C.c_int, C.POINTER(Patch)]
        _lib.so_patch_cells.argtypes = [C.POINTER(Cfg), C.POINTER(Patch), C.POINTER(C.c_int)]
        _lib.so_num_patches.argtypes = [C.POINTER(Cfg)]
        _lib.so_set_svd_mode.argtypes = [C.c_int]
        _lib.so_patch_pt.argtypes = [C.POINTER(Cfg), C.c_int, dp]
        _lib.so_set_solver_noise.argtypes = [C.c_double, C.c_ulonglong]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_cfg(nref=0, n_sub=2, oversampling=1, spacedim=1, stabilize=1, reuse_full=0, proj_quirk=0,
             n_cells=0):
    return Cfg(nref, n_cells, n_sub, oversampling, spacedim, stabilize, reuse_full, proj_quirk)


def n_cells_per_side(cfg):
    return cfg.n_cells if cfg.n_cells > 0 else 1 << cfg.nref


def num_patches(cfg):
    return lib().so_num_patches(C.byref(cfg))


def patch_info(cfg, pid):
    p = Patch()
    lib().so_patch_init(C.byref(cfg), pid, C.byref(p))
    return p


def patch_cells(cfg, pid):
    p = patch_info(cfg, pid)
    buf = (C.c_int * (p.mx * p.my))()
    n = lib().so_patch_cells(C.byref(cfg), C.byref(p), buf)
    return list(buf)[:n]


def _coef_ptrs(coefs):
    arrs = [np.ascontiguousarray(c, dtype=np.float64) for c in coefs]
    ptrs = (C.POINTER(C.c_double) * 2)()
    for i, a in enumerate(arrs):
        ptrs[i] = _dp(a)
    return arrs, ptrs


def fill_coefficient(seed, dist, lo, hi, n_elems):
    out = np.empty(n_elems * n_elems * 4, dtype=np.float64)
    lib().so_fill_coefficient(seed, dist, lo, hi, n_elems, _dp(out))
    return out


def fill_coefficient_rand(lo, hi, r, n_elems, seed=1):
    """Reference-like Alpha(lo,hi,r) field (Diffusion.h:19-51) using glibc rand()."""
    libc = C.CDLL("libc.so.6")
    libc.srand(seed)
    out = np.empty(n_elems * n_elems * 4, dtype=np.float64)
    lib().so_fill_coefficient_rand(lo, hi, r, n_elems, _dp(out))
    return out


def patch_basis(cfg, coefs, pid):
    """-> (phi[s, n_f], psi[s, n_f], Diag)."""
    p = patch_info(cfg, pid)
    s = cfg.spacedim
    phi = np.zeros((s, p.n_f))
    psi = np.zeros((s, p.n_f))
    d = Diag()
    keep, ptrs = _coef_ptrs(coefs)
    rc = lib().so_patch_basis(C.byref(cfg), ptrs, pid, _dp(phi), _dp(psi), C.byref(d))
    if rc:
        raise RuntimeError("so_patch_basis failed rc=%d" % rc)
    return phi, psi, d


def patch_debug(cfg, coefs, pid):
    p = patch_info(cfg, pid)
    M = np.zeros((p.n_c, p.n_c))
    D = np.zeros((p.n_c, p.n_c))
    BD = np.zeros((max(p.n_b, 1), p.n_c))
    X = np.zeros((p.n_f, p.n_c))
    bd = (C.c_int * max(p.n_b, 1))()
    keep, ptrs = _coef_ptrs(coefs)
    rc = lib().so_patch_debug(C.byref(cfg), ptrs, pid, _dp(M), _dp(D), _dp(BD), bd, _dp(X))
    if rc:
        raise RuntimeError("so_patch_debug failed rc=%d" % rc)
    return dict(M=M, D=D, BD=BD[:p.n_b], bdofs=np.array(list(bd)[:p.n_b]), X=X)


def basis_many(cfg, coefs, ids, offsets, total, nthreads=1):
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    phi = np.zeros(total)
    psi = np.zeros(total)
    keep, ptrs = _coef_ptrs(coefs)
    rc = lib().so_basis_many(C.byref(cfg), ptrs, ids.ctypes.data_as(C.POINTER(C.c_int)), len(ids),
                             _dp(phi), _dp(psi), offsets.ctypes.data_as(C.POINTER(C.c_longlong)),
                             nthreads)
    if rc:
        raise RuntimeError("so_basis_many failed rc=%d" % rc)
    return phi, psi


def assemble_patch(cfg, coefs, pid):
    """-> stencil[n_nodes, 9, s, s]"""
    p = patch_info(cfg, pid)
    s = cfg.spacedim
    st = np.zeros((p.n_f // s, 9, s, s))
    keep, ptrs = _coef_ptrs(coefs)
    lib().so_assemble_patch(C.byref(cfg), C.byref(p), ptrs, _dp(st))
    return st


def solve_interior(nx, ny, s, stencil, rhs):
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    out = np.zeros_like(rhs)
    st = np.ascontiguousarray(stencil, dtype=np.float64)
    rc = lib().so_solve_interior(nx, ny, s, _dp(st), _dp(rhs), rhs.shape[1], _dp(out))
    if rc:
        raise RuntimeError("so_solve_interior failed rc=%d" % rc)
    return out


def local_matrix_poisson(alpha):
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    K = np.zeros((4, 4))
    lib().so_local_matrix_poisson(_dp(a), _dp(K))
    return K


def local_matrix_elasticity(lam, mu):
    l_ = np.ascontiguousarray(lam, dtype=np.float64)
    m_ = np.ascontiguousarray(mu, dtype=np.float64)
    K = np.zeros((8, 8))
    lib().so_local_matrix_elasticity(_dp(l_), _dp(m_), _dp(K))
    return K


def fe_q_iso_q1_cell_matrix(dim, n):
    nd = (n + 1) ** dim
    M = np.zeros((nd, nd))
    lib().so_fe_q_iso_q1_cell_matrix(dim, n, _dp(M))
    return M


def set_svd_mode(mode):
    lib().so_set_svd_mode(mode)


def set_solver_noise(eps, seed=0):
    """Conditioning probe: relative noise eps on X before the selection stage (0 = off)."""
    lib().so_set_solver_noise(eps, seed)


def selection_conditioning(cfg, coefs, pid, eps=1e-13):
    """How far phi and the decisions of patch `pid` move when X carries the rounding noise of
    another fp64 solver (two noise draws).  -> (max |dphi|, decisions_stable)."""
    phi0, _, d0 = patch_basis(cfg, coefs, pid)
    spread, stable = 0.0, True
    try:
        for seed in (1, 2):
            set_solver_noise(eps, seed)
            phi1, _, d1 = patch_basis(cfg, coefs, pid)
            spread = max(spread, float(np.abs(phi1 - phi0).max()))
            for c in range(cfg.spacedim):
                stable = stable and d1.n_cut[c] == d0.n_cut[c] and d1.n_dropped[c] == d0.n_dropped[c]
    finally:
        set_solver_noise(0.0, 0)
    return spread, stable


def patch_pt(cfg, pid):
    p = patch_info(cfg, pid)
    PT = np.zeros((p.n_f, p.n_c))
    lib().so_patch_pt(C.byref(cfg), pid, _dp(PT))
    return PT
