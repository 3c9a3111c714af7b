# Timing experiment: phases of the selection stage (k_select / fused select_patch) per patch, diag build.
# usage: python tools/select_timeline.py C2|C4
import ctypes as C
import os
import sys

os.environ["SLOD_DIAG"] = str(1 << 20)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SLOD_LIB_PATH", os.path.join(ROOT, "dealii-slod_amd", "lib", "libslod_hip_diag.so"))
os.environ["SLOD_BALANCE"] = "0"
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
import numpy as np
import torch
import slod_amd
from slod_amd.synthetic import fill_coefficient

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C4"
s = 2 if cfgname == "C4" else 1
g = slod_amd.Slod(device=0, nref=5, n_sub=8, oversampling=2, spacedim=s, stabilize=1)
dev = torch.device("cuda", 0)
for f in range(s):
    t = torch.from_numpy(fill_coefficient(20250614 + f, "D1e4", g.NE)).to(dev)
    g.set_coefficient_device(f, t.data_ptr(), t.numel())
ids = np.arange(g.num_patches, dtype=np.uint32)
plan = g.plan(ids)
basis = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev)
premult = torch.zeros_like(basis)
for _ in range(3):
    plan.execute(basis.data_ptr(), premult.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ncm = 25 * s
buf = np.zeros(len(ids) * ncm * ncm)
lib = g.lib
lib.slod_debug_read_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
assert lib.slod_debug_read_ms(plan.p, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
r = buf.reshape(len(ids), ncm * ncm)[:, :12] / 100.0
print("kernel_ms", plan.kernel_ms())
full = [k for k, pid in enumerate(ids) if (lambda i: i.mx == 5 and i.my == 5)(g.patch_layout(int(pid)))]
rr = r[full]
ph = dict(D=rr[:, 4] - rr[:, 3], fill=rr[:, 5], mult=rr[:, 6], qr=rr[:, 7], rinv=rr[:, 8] - rr[:, 4] - rr[:, 5] - rr[:, 6] - rr[:, 7],
          svd=rr[:, 9] - rr[:, 8], phi=rr[:, 10] - rr[:, 9], psi=rr[:, 11] - rr[:, 10], total_after_M=rr[:, 11] - rr[:, 3])
slow = ph["svd"] > np.median(ph["svd"]) * 3 + 20
for name, m in (("fast path", ~slow), ("SVD fallback", slow)):
    if m.sum():
        print("%-13s n=%3d " % (name, m.sum()) + "  ".join("%s %.1f" % (k, v[m].mean()) for k, v in ph.items()))
