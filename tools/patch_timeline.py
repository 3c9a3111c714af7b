# Timing experiment: per-patch timeline of the fused solve+select launch on C2 (SLOD_DIAG bit 20
# makes thread 0 of every workgroup stamp the 100 MHz wall clock at start / after the solve /
# after the selection stage).  Prints, per patch shape, start offset and the two durations.
import ctypes as C
import os
import sys

os.environ["SLOD_DIAG"] = str(1 << 20)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
import numpy as np
import torch
import slod_amd
from slod_amd.synthetic import fill_coefficient

g = slod_amd.Slod(device=0, nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
dev = torch.device("cuda", 0)
t = torch.from_numpy(fill_coefficient(20250614, "D1e4", g.NE)).to(dev)
g.set_coefficient_device(0, t.data_ptr(), t.numel())
ids = np.arange(g.num_patches, dtype=np.uint32)
if len(sys.argv) > 1:  # only the first N full (5x5) patches: N = 256 gives one workgroup per CU
    full = [int(i) for i in ids if g.patch_layout(int(i)).mx == 5 and g.patch_layout(int(i)).my == 5]
    ids = np.array(full[:int(sys.argv[1])], dtype=np.uint32)
plan = g.plan(ids)
basis = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
premult = torch.zeros_like(basis)
for _ in range(3):
    plan.execute(basis.data_ptr(), premult.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ncm = 25
buf = np.zeros(len(ids) * ncm * ncm)
lib = g.lib
lib.slod_debug_read_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
assert lib.slod_debug_read_ms(plan.p, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
raw = buf.reshape(len(ids), ncm * ncm)[:, :12] / 100.0  # microseconds
tt = raw[:, :3].copy()
if not os.environ.get('SLOD_FUSE_SELECT', '1') != '0':
    tt[:, 2] = raw[:, 11]
t0 = tt[:, 0].min()
shapes = {}
for k, pid in enumerate(ids):
    info = g.patch_layout(int(pid))
    key = (min(info.mx, info.my), max(info.mx, info.my))
    shapes.setdefault(key, []).append((tt[k, 0] - t0, tt[k, 1] - tt[k, 0], tt[k, 2] - tt[k, 1], tt[k, 2] - t0))
print("launch span %.1f us" % (tt[:, 2].max() - t0))
for key in sorted(shapes):
    a = np.array(shapes[key])
    print("shape %dx%d  n=%4d  start %6.1f..%6.1f  solve mean %6.1f max %6.1f  select mean %6.1f max %6.1f  end max %6.1f"
          % (key[0], key[1], len(a), a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].max(), a[:, 2].mean(),
             a[:, 2].max(), a[:, 3].max()))
full = np.array(shapes[(5, 5)])
print("5x5 select histogram (us):", np.histogram(full[:, 2], bins=8))
print("5x5 solve histogram (us):", np.histogram(full[:, 1], bins=8))

# phases of the selection stage (full patches), fast path vs SVD fallback
sel = [k for k, pid in enumerate(ids) if (lambda i: i.mx == 5 and i.my == 5)(g.patch_layout(int(pid)))]
r = raw[sel]
t_end_solve = r[:, 1]
ph = dict(M=r[:, 3] - t_end_solve, D=r[:, 4] - r[:, 3], fill=r[:, 5], mult=r[:, 6], qr=r[:, 7],
          rinv=r[:, 8] - r[:, 4] - r[:, 5] - r[:, 6] - r[:, 7], svd=r[:, 9] - r[:, 8], phi=r[:, 10] - r[:, 9],
          psi=r[:, 11] - r[:, 10])
slow = ph["svd"] > 20.0
for name, m in (("fast path", ~slow), ("SVD fallback", slow)):
    print("%-13s n=%3d " % (name, m.sum()) + "  ".join("%s %.1f" % (k, v[m].mean()) for k, v in ph.items()))

# placement: which patches share a CU (HW_ID: CU_ID bits 11:8, SH_ID 12, SE_ID 15:13; XCC_ID bits 3:0)
hw = buf.reshape(len(ids), ncm * ncm)[:, 12].astype(np.int64)
xcc = buf.reshape(len(ids), ncm * ncm)[:, 13].astype(np.int64) & 15
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
groups = {}
for k in range(len(ids)):
    groups.setdefault(int(cu[k]), []).append(k)
sizes = np.array([len(v) for v in groups.values()])
print("distinct CUs %d, workgroups per CU: min %d max %d" % (len(groups), sizes.min(), sizes.max()))
for key in list(sorted(groups))[:6]:
    print("  cu %5d: blocks %s" % (key, groups[key]))
end = tt[:, 2] - t0
per_cu_end = np.array([end[v].max() for v in groups.values()])
print("per-CU finish time (us): mean %.1f  min %.1f  max %.1f" % (per_cu_end.mean(), per_cu_end.min(), per_cu_end.max()))
work = (tt[:, 2] - tt[:, 0])
print("sum of patch times per CU: mean %.1f max %.1f" % (np.mean([work[v].sum() for v in groups.values()]), np.max([work[v].sum() for v in groups.values()])))

# wave -> SIMD placement (HW_ID bits 5:4) of the four roles: GJ chain 0/1 (roles 0,1), helpers (2,3)
if ncm * ncm >= 24:
    simd = (buf.reshape(len(ids), ncm * ncm)[:, 20:24].astype(np.int64) >> 4) & 3
    import collections
    print("role->SIMD patterns:", collections.Counter(tuple(r) for r in simd).most_common(6))
    for key in list(sorted(groups))[:4]:
        print("  cu %5d: role SIMDs per workgroup %s" % (key, [tuple(simd[k]) for k in groups[key]]))
    gj = np.zeros((len(groups), 4), int)
    for i, v in enumerate(groups.values()):
        for k in v:
            gj[i, simd[k, 0]] += 1
            gj[i, simd[k, 1]] += 1
    print("GJ waves per SIMD (per CU): histogram of max", np.bincount(gj.max(axis=1)))
