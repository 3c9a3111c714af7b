"""ctypes binding of libslod_hip.so (include/slod.h) for tests and bench.py.

This is plumbing only: every compute call goes through the C-ABI into the HIP kernels.
There is no CPU fallback -- loading fails loudly if the library has not been built, and
compute entry points raise SlodError if no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libslod_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "slod.h")


class SlodError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slod error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "dim", "spacedim", "n_global_refinements", "n_cells_per_side", "n_subdivisions",
        "oversampling", "lod_stabilization", "constant_coefficients", "projection_quirk",
        "n_problems", "device", "reserved")]


class PatchInfo(C.Structure):
    _fields_ = [("cx", C.c_int32), ("cy", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("mx", C.c_int32), ("my", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32),
                ("side_domain", C.c_int32 * 4), ("n_fine", C.c_int32), ("n_internal", C.c_int32),
                ("n_boundary", C.c_int32), ("n_coarse", C.c_int32), ("is_lod", C.c_int32)]


class PatchDiag(C.Structure):
    """slod_patch_diag: decisions of the SLOD selection stage for one (patch, component)."""
    _fields_ = [("path", C.c_int32), ("n_cut", C.c_int32), ("n_dropped", C.c_int32), ("sweeps", C.c_int32),
                ("dinf", C.c_double), ("sigma_max", C.c_double), ("sigma_min", C.c_double)]


_lib = None


def load():
    """Load libslod_hip.so; raises OSError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # tools/ may point at the timing-experiment build (lib/libslod_hip_diag.so, `make diag`)
    path = os.environ.get("SLOD_LIB_PATH", LIB_PATH)
    if not os.path.exists(path):
        raise OSError("%s not built: run `make -C dealii-slod_amd` "
                      "(or __graft_entry__.build()); there is no CPU fallback" % path)
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (soname libamdhip64.so.7,
    # the same as /opt/rocm's).  If torch is imported AFTER this library, a second runtime
    # gets loaded and whichever initialises second sees no device.  Importing torch first
    # makes libslod_hip bind to torch's copy (torch is only plumbing here: device memory,
    # streams, torch.distributed).  Non-Python users link /opt/rocm's runtime as usual.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, dp, u32p, u64p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.slod_abi_version.restype = C.c_int
    lib.slod_last_error.restype = C.c_char_p
    lib.slod_last_error.argtypes = [vp]
    lib.slod_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.slod_destroy.argtypes = [vp]
    lib.slod_destroy.restype = None
    lib.slod_num_patches.argtypes = [vp]
    lib.slod_patch_layout.argtypes = [vp, C.c_uint32, C.POINTER(PatchInfo)]
    lib.slod_patch_cells.argtypes = [vp, C.c_uint32, u32p, C.c_size_t]
    lib.slod_patch_dof_permutation.argtypes = [vp, C.c_uint32, u32p, C.c_size_t]
    lib.slod_partition.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]
    lib.slod_set_coefficient.argtypes = [vp, C.c_uint32, C.c_int, vp, C.c_int, C.c_size_t, C.c_int]
    lib.slod_plan_create.argtypes = [vp, u32p, C.c_size_t, u64p, C.POINTER(vp)]
    lib.slod_plan_destroy.argtypes = [vp]
    lib.slod_plan_destroy.restype = None
    lib.slod_plan_stride.argtypes = [vp]
    lib.slod_plan_stride.restype = C.c_size_t
    lib.slod_plan_output_size.argtypes = [vp]
    lib.slod_plan_output_size.restype = C.c_size_t
    lib.slod_plan_execute.argtypes = [vp, vp, vp, vp]
    lib.slod_plan_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.slod_plan_profile.argtypes = [vp, C.c_int]
    lib.slod_plan_status.argtypes = [vp]
    lib.slod_plan_patch_layout.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(PatchInfo), u32p]
    lib.slod_plan_set_overlap.argtypes = [vp, C.c_int]
    lib.slod_plan_join.argtypes = [vp, vp]
    lib.slod_plan_diagnostics.argtypes = [vp, C.POINTER(PatchDiag), C.c_size_t]
    lib.slod_compute_basis.argtypes = [vp, u32p, C.c_size_t, dp, dp, u64p]
    lib.slod_comm_last_error.restype = C.c_char_p
    lib.slod_comm_last_error.argtypes = [vp]
    lib.slod_comm_unique_id.argtypes = [C.c_char_p]
    lib.slod_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.slod_comm_destroy.argtypes = [vp]
    lib.slod_comm_destroy.restype = None
    lib.slod_comm_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.slod_gather_piece.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]
    lib.slod_plan_execute_allgather.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp]
    lib.slod_lod_row_capacity.argtypes = [vp]
    lib.slod_lod_pattern.argtypes = [vp, C.c_uint32, u32p, C.c_size_t]
    lib.slod_lod_matrix.argtypes = [vp, u32p, C.c_size_t, vp, vp, C.c_size_t, vp, vp, vp]
    lib.slod_lod_rhs.argtypes = [vp, u32p, C.c_size_t, vp, C.c_size_t, vp, vp, vp]
    lib.slod_lod_solve.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_int, dp]
    lib.slod_lod_reconstruct.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    lib.slod_fem_rhs.argtypes = [vp, vp, vp, vp]
    lib.slod_fem_solve.argtypes = [vp, C.c_uint32, vp, vp, C.c_double, C.c_int, dp]
    lib.slod_device_patch_layout.argtypes = [vp, u32p, C.c_size_t, C.POINTER(PatchInfo)]
    lib.slod_sample_coefficient.argtypes = [vp, C.c_uint32, C.c_int, vp, C.c_int]
    lib.slod_assemble_stiffness_for_patch.argtypes = [vp, C.c_uint32, dp]
    lib.slod_patch_solution.argtypes = [vp, C.c_uint32, dp]
    _lib = lib
    return lib


def declared_symbols():
    """Entry points declared in include/slod.h (used by the symbol-export test)."""
    import re
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(slod_[a-z_0-9]+)\s*\(", txt)))


def partition(n_total, n_ranks, rank):
    b, e = C.c_uint64(), C.c_uint64()
    rc = load().slod_partition(n_total, n_ranks, rank, C.byref(b), C.byref(e))
    if rc:
        raise SlodError(rc, "slod_partition: bad arguments")
    return b.value, e.value


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def gather_piece(patches_per_rank, n_pieces, piece):
    f, c = C.c_uint64(), C.c_uint64()
    rc = load().slod_gather_piece(patches_per_rank, n_pieces, piece, C.byref(f), C.byref(c))
    if rc:
        raise SlodError(rc, "slod_gather_piece: bad arguments")
    return f.value, c.value


class Comm:
    """slod_comm: RCCL communicator of the C-ABI (what a C++ host uses; bench.py's N > 1 path goes
    through torch.distributed instead)."""

    def __init__(self, comm_id, n_ranks, rank, device=0):
        self.lib = load()
        self.c = C.c_void_p()
        rc = self.lib.slod_comm_create(comm_id, n_ranks, rank, device, C.byref(self.c))
        if rc:
            raise SlodError(rc, self.lib.slod_comm_last_error(None).decode())
        self.n_ranks, self.rank = n_ranks, rank

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = load().slod_comm_unique_id(buf)
        if rc:
            raise SlodError(rc, load().slod_comm_last_error(None).decode())
        return buf.raw

    def allgather(self, d_send, d_recv, count, stream):
        rc = self.lib.slod_comm_allgather(self.c, d_send, d_recv, count, stream)
        if rc:
            raise SlodError(rc, self.lib.slod_comm_last_error(self.c).decode())

    def close(self):
        if self.c:
            self.lib.slod_comm_destroy(self.c)
            self.c = None


class Plan:
    def __init__(self, slod, gids, offsets=None):
        self.slod = slod
        self.lib = slod.lib
        self.gids = np.ascontiguousarray(gids, dtype=np.uint32)
        off_p = None
        if offsets is not None:
            self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            off_p = self.offsets.ctypes.data_as(C.POINTER(C.c_uint64))
        self.p = C.c_void_p()
        rc = self.lib.slod_plan_create(slod.h, self.gids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       len(self.gids), off_p, C.byref(self.p))
        slod._check(rc)
        self.stride = self.lib.slod_plan_stride(self.p)
        self.output_size = self.lib.slod_plan_output_size(self.p)

    def execute(self, d_basis_ptr, d_premult_ptr, stream_ptr=None):
        """Asynchronous launch; arguments are raw device pointers (ints)."""
        self.slod._check(self.lib.slod_plan_execute(self.p, d_basis_ptr, d_premult_ptr, stream_ptr))

    def execute_allgather(self, comm, d_basis_all, d_premult_all, patches_per_rank, n_pieces, compute_stream,
                          comm_stream):
        self.slod._check(self.lib.slod_plan_execute_allgather(self.p, comm.c, d_basis_all, d_premult_all,
                                                              patches_per_rank, n_pieces, compute_stream, comm_stream))

    def profile(self, depth):
        self.slod._check(self.lib.slod_plan_profile(self.p, depth))

    def kernel_ms(self):
        ms = (C.c_float * 3)()
        self.slod._check(self.lib.slod_plan_kernel_ms(self.p, ms))
        return [float(x) for x in ms]

    def status(self):
        self.slod._check(self.lib.slod_plan_status(self.p))

    def set_overlap(self, depth):
        """depth 2: consecutive executes overlap on two internal streams (join() / status() order the caller after them)."""
        self.slod._check(self.lib.slod_plan_set_overlap(self.p, depth))

    def join(self, stream_ptr=None):
        self.slod._check(self.lib.slod_plan_join(self.p, stream_ptr))

    def patch_layout(self, k, launch_order=False):
        """(PatchInfo, plan_index) of the k-th descriptor the kernels launch with (device read-back)."""
        info, idx = PatchInfo(), C.c_uint32()
        self.slod._check(self.lib.slod_plan_patch_layout(self.p, k, 1 if launch_order else 0, C.byref(info), C.byref(idx)))
        return info, idx.value

    def diagnostics(self):
        """[(patch k, component d)] -> PatchDiag of the last execute."""
        n = len(self.gids) * self.slod.spacedim
        buf = (PatchDiag * max(n, 1))()
        rc = self.lib.slod_plan_diagnostics(self.p, buf, n)
        if rc < 0:
            self.slod._check(rc)
        return [buf[i] for i in range(n)]

    def close(self):
        if self.p:
            self.lib.slod_plan_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Slod:
    """Thin OO wrapper over a slod_handle."""

    def __init__(self, nref=0, n_sub=2, oversampling=1, spacedim=1, stabilize=1, reuse_full=0,
                 proj_quirk=0, n_cells=0, n_problems=1, device=0):
        self.lib = load()
        self.cfg = Config(2, spacedim, nref, n_cells, n_sub, oversampling, stabilize, reuse_full,
                          proj_quirk, n_problems, device, 0)
        self.h = C.c_void_p()
        rc = self.lib.slod_create(C.byref(self.cfg), C.byref(self.h))
        if rc:
            raise SlodError(rc, self.lib.slod_last_error(None).decode())
        self.N = n_cells if n_cells > 0 else 1 << nref
        self.NE = self.N * n_sub
        self.spacedim = spacedim
        self.num_patches = self.lib.slod_num_patches(self.h)

    def _check(self, rc):
        if rc:
            raise SlodError(rc, self.lib.slod_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.slod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def patch_layout(self, pid):
        info = PatchInfo()
        self._check(self.lib.slod_patch_layout(self.h, pid, C.byref(info)))
        return info

    def patch_cells(self, pid):
        info = self.patch_layout(pid)
        buf = (C.c_uint32 * (info.mx * info.my))()
        n = self.lib.slod_patch_cells(self.h, pid, buf, len(buf))
        if n < 0:
            self._check(n)
        return list(buf)[:n]

    def patch_dof_permutation(self, pid):
        info = self.patch_layout(pid)
        buf = (C.c_uint32 * info.n_fine)()
        n = self.lib.slod_patch_dof_permutation(self.h, pid, buf, len(buf))
        if n < 0:
            self._check(n)
        return np.array(buf[:n], dtype=np.int64)

    def set_coefficient(self, field, data, problem=0, per_qp=True):
        a = np.ascontiguousarray(data, dtype=np.float64).ravel()
        self._check(self.lib.slod_set_coefficient(self.h, problem, field, a.ctypes.data, 1 if per_qp else 0,
                                                  a.size, 0))

    def set_coefficient_device(self, field, dev_ptr, count, problem=0, per_qp=True):
        self._check(self.lib.slod_set_coefficient(self.h, problem, field, dev_ptr, 1 if per_qp else 0,
                                                  count, 1))

    def plan(self, gids, offsets=None):
        return Plan(self, gids, offsets)

    # ---- consumers of (phi, psi): the global LOD system (raw device pointers as ints) ----
    def lod_row_capacity(self):
        return self.lib.slod_lod_row_capacity(self.h)

    def lod_pattern(self, pid):
        buf = (C.c_uint32 * self.lod_row_capacity())()
        n = self.lib.slod_lod_pattern(self.h, pid, buf, len(buf))
        if n < 0:
            self._check(n)
        return list(buf)[:n]

    def lod_matrix(self, rows, d_basis, d_premult, stride, d_values, d_cols, stream=None):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        self._check(self.lib.slod_lod_matrix(self.h, rows.ctypes.data_as(C.POINTER(C.c_uint32)), len(rows), d_basis,
                                             d_premult, stride, d_values, d_cols, stream))

    def lod_rhs(self, rows, d_basis, stride, d_fine_rhs, d_out, stream=None):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        self._check(self.lib.slod_lod_rhs(self.h, rows.ctypes.data_as(C.POINTER(C.c_uint32)), len(rows), d_basis, stride,
                                          d_fine_rhs, d_out, stream))

    def lod_solve(self, d_values, d_cols, d_rhs, d_u, rel_tol=1e-12, max_iterations=2000):
        res = C.c_double()
        it = self.lib.slod_lod_solve(self.h, d_values, d_cols, d_rhs, d_u, rel_tol, max_iterations, C.byref(res))
        if it < 0:
            self._check(it)
        return it, res.value

    def lod_reconstruct(self, d_basis, stride, d_u, d_fine, stream=None):
        self._check(self.lib.slod_lod_reconstruct(self.h, d_basis, stride, d_u, d_fine, stream))

    def fem_rhs(self, d_f_qp, d_fine_rhs, stream=None):
        """Fine FEM load vector (d_f_qp = None: f = 1)."""
        self._check(self.lib.slod_fem_rhs(self.h, d_f_qp, d_fine_rhs, stream))

    def fem_solve(self, d_fine_rhs, d_fine_u, rel_tol=1e-12, max_iterations=20000, problem=0):
        res = C.c_double()
        it = self.lib.slod_fem_solve(self.h, problem, d_fine_rhs, d_fine_u, rel_tol, max_iterations, C.byref(res))
        if it < 0:
            self._check(it)
        return it, res.value

    def device_patch_layout(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = (PatchInfo * max(len(ids), 1))()
        self._check(self.lib.slod_device_patch_layout(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint32)), len(ids), out))
        return [out[i] for i in range(len(ids))]

    def sample_coefficient(self, field, d_vals, r, problem=0):
        self._check(self.lib.slod_sample_coefficient(self.h, problem, field, d_vals, r))

    def compute_basis(self, gids, offsets=None, total=None):
        """Host-buffer path (slod_compute_basis). Returns (basis, premult) flat arrays."""
        gids = np.ascontiguousarray(gids, dtype=np.uint32)
        s = self.spacedim
        if offsets is None:
            sizes = [s * self.patch_layout(int(g) % self.num_patches).n_fine for g in gids]
            offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64) if len(gids) else \
                np.zeros(0, np.uint64)
            total = int(np.sum(sizes))
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        basis = np.zeros(total)
        premult = np.zeros(total)
        self._check(self.lib.slod_compute_basis(
            self.h, gids.ctypes.data_as(C.POINTER(C.c_uint32)), len(gids), _dp(basis), _dp(premult),
            offsets.ctypes.data_as(C.POINTER(C.c_uint64))))
        return basis, premult, offsets

    def assemble_stiffness_for_patch(self, gid):
        info = self.patch_layout(gid % self.num_patches)
        s = self.spacedim
        st = np.zeros((info.n_fine // s, 9, s, s))
        self._check(self.lib.slod_assemble_stiffness_for_patch(self.h, gid, _dp(st)))
        return st

    def patch_solution(self, gid):
        info = self.patch_layout(gid % self.num_patches)
        X = np.zeros((info.n_fine, info.n_coarse))
        self._check(self.lib.slod_patch_solution(self.h, gid, _dp(X)))
        return X
