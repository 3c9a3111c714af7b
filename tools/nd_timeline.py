# Timing experiment: per-patch phase timeline of k_solve_nd on C2 (diag build: make -C dealii-slod_amd diag;
# SLOD_LIB_PATH=dealii-slod_amd/lib/libslod_hip_diag.so).  Thread 0 of every workgroup stamps the 100 MHz
# wall clock at the phase boundaries (SLOD_DIAG bit 20).
import ctypes as C
import os
import sys

os.environ["SLOD_DIAG"] = str(1 << 20)
os.environ.setdefault("SLOD_SOLVE", "nd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SLOD_LIB_PATH", os.path.join(ROOT, "dealii-slod_amd", "lib", "libslod_hip_diag.so"))
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
if len(sys.argv) > 1:  # only the first N full (5x5) patches
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
raw = buf.reshape(len(ids), ncm * ncm) / 100.0  # microseconds
st = raw[:, 32:46]
names = ["assemble", "factor", "condense", "rhs", "skeleton", "-", "H:bwd", "back E", "back cells"]
# stamps: 0 start, 1 after assemble, 2 after factor, 3 after condense, 4 after rhs, 6 after the skeleton sweep,
# 12 before the backward line sweep, 7 after it, 8 after the edge back substitution, 9 end
order = [0, 1, 2, 3, 4, 6, 12, 7, 8, 9]
t0 = st[:, 0].min()
sel_end = st[:, 9]  # (selection stage: its own launch)
shapes = {}
for k, pid in enumerate(ids):
    info = g.patch_layout(int(pid))
    shapes.setdefault((min(info.mx, info.my), max(info.mx, info.my)), []).append(k)
print("launch span %.1f us (kernel_ms %s)" % (max(sel_end.max(), st[:, 9].max()) - t0, plan.kernel_ms()))
for key in sorted(shapes):
    idx = shapes[key]
    d = np.array([[st[k, order[j + 1]] - st[k, order[j]] for j in range(len(order) - 1)] for k in idx])
    tot = st[idx, 9] - st[idx, 0]
    sel = sel_end[idx] - st[idx, 9]
    print("shape %dx%d n=%4d solve %.1f (max %.1f) select %.1f (max %.1f) | " % (key[0], key[1], len(idx), tot.mean(), tot.max(),
                                                                           sel.mean(), sel.max())
          + "  ".join("%s %.1f" % (names[j], d[:, j].mean()) for j in range(len(order) - 1)))
