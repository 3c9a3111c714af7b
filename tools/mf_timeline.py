# Timing experiment (diag build): per-wave phase cycles of k_solve_mf on C2, averaged over patches.
#   GJ waves (0,1):     0 wait/pre, 1 sweep, 2 V store, 3 Schur update, 4 barrier
#   helper waves (2,3): 0 pre, 1 RHS operand, 2 GEMM + store, 3 band fetch, 4 barrier
#   all waves, backward: 5 MFMA+store of the previous line (and loop entry), 6 strip/Z/band/operand, 7 tail
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SLOD_DIAG"] = str(1 << 21)
os.environ.setdefault("SLOD_LIB_PATH", os.path.join(ROOT, "dealii-slod_amd", "lib", "libslod_hip_diag.so"))
os.environ.setdefault("SLOD_SOLVE", "mf")
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
import numpy as np, torch, slod_amd
from slod_amd.synthetic import fill_coefficient
g = slod_amd.Slod(device=0, nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
dev = torch.device("cuda", 0)
t = torch.from_numpy(fill_coefficient(20250614, "D1e4", g.NE)).to(dev)
g.set_coefficient_device(0, t.data_ptr(), t.numel())
ids = np.arange(g.num_patches, dtype=np.uint32)
if len(sys.argv) > 1:
    full = [int(i) for i in ids if g.patch_layout(int(i)).mx == 5 and g.patch_layout(int(i)).my == 5]
    ids = np.array(full[:int(sys.argv[1])], dtype=np.uint32)
plan = g.plan(ids)
basis = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev); premult = torch.zeros_like(basis)
for _ in range(3):
    plan.execute(basis.data_ptr(), premult.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("kernel ms", plan.kernel_ms())
ncm = 25
buf = np.zeros(len(ids) * ncm * ncm)
lib = g.lib
lib.slod_debug_read_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
assert lib.slod_debug_read_ms(plan.p, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
raw = buf.reshape(len(ids), ncm * ncm)[:, 16:48].reshape(len(ids), 4, 8)
full = np.array([g.patch_layout(int(i)).mx == 5 and g.patch_layout(int(i)).my == 5 for i in ids])
for name, sel in (("full patches", full), ("rim patches", ~full)):
    if sel.sum() == 0:
        continue
    print(name, int(sel.sum()))
    for w in range(4):
        print("  wave %d kcycles: %s   total %.0f" % (w, " ".join("%7.1f" % (x / 1e3) for x in raw[sel, w].mean(axis=0)), raw[sel, w].mean(axis=0).sum() / 1e3))
