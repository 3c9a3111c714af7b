# Timing experiment: phases of k_solve_tw per patch (diag build; thread 0 = Gauss-Jordan wave of chain 0).
import ctypes as C, os, sys
os.environ["SLOD_DIAG"] = str((1 << 20) | int(os.environ.get("TW_EXTRA_DIAG", "0")))
os.environ["SLOD_BALANCE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SLOD_LIB_PATH", os.path.join(ROOT, "dealii-slod_amd", "lib", "libslod_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
import numpy as np, torch, slod_amd
from slod_amd.synthetic import fill_coefficient
g = slod_amd.Slod(device=0, nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
dev = torch.device("cuda", 0)
t = torch.from_numpy(fill_coefficient(20250614, "D1e4", g.NE)).to(dev)
g.set_coefficient_device(0, t.data_ptr(), t.numel())
ids = np.arange(g.num_patches, dtype=np.uint32)
plan = g.plan(ids)
b = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev); q = torch.zeros_like(b)
for _ in range(3):
    plan.execute(b.data_ptr(), q.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ncm = 25
buf = np.zeros(len(ids) * ncm * ncm)
g.lib.slod_debug_read_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
assert g.lib.slod_debug_read_ms(plan.p, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
r = buf.reshape(len(ids), ncm * ncm) / 100.0
full = [k for k, pid in enumerate(ids) if (lambda i: i.mx == 5 and i.my == 5)(g.patch_layout(int(pid)))]
st = r[full][:, 32:36]; end_solve = r[full][:, 1]; end_sel = r[full][:, 2]
print("kernel_ms", plan.kernel_ms())
print("5x5 patches: assemble+prologue %.1f  forward sweep %.1f  meeting line %.1f  backward %.1f  select %.1f  total %.1f" % (
    (st[:, 1] - st[:, 0]).mean(), (st[:, 2] - st[:, 1]).mean(), (st[:, 3] - st[:, 2]).mean(), (end_solve - st[:, 3]).mean(),
    (end_sel - end_solve).mean(), (end_sel - st[:, 0]).mean()))
x = r[full]
print("forward sweep, sums over the steps (us): GJ wave (chain 0): sweep %.1f  store+next_S %.1f  barrier wait %.1f | helper (chain 0): R+Z %.1f  bands %.1f  barrier wait %.1f" % (
    x[:, 40].mean(), x[:, 41].mean(), x[:, 42].mean(), x[:, 44].mean(), x[:, 45].mean(), x[:, 46].mean()))
print("  of store+next_S: the V store alone %.1f us" % x[:, 43].mean())
print("  of R+Z: building the RHS block alone %.1f us" % x[:, 47].mean())
hw = r[:, 20:24] * 100.0
simd = (hw.astype(np.int64) >> 4) & 3
import collections
print("SIMD of waves (GJ0, GJ1, helper0, helper1), most common placements:", collections.Counter(map(tuple, simd.tolist())).most_common(6))
# which workgroups share a CU: (xcc, se, sh, cu) key from wave 0's HW_ID and XCC_ID
hw0 = (r[:, 12] * 100.0).astype(np.int64); xcc = (r[:, 13] * 100.0).astype(np.int64) & 15
key = [(int(x), int(h >> 13) & 7, int(h >> 12) & 1, int(h >> 8) & 15) for h, x in zip(hw0, xcc)]
per_cu = collections.defaultdict(list)
for k, (kk, sd) in enumerate(zip(key, simd.tolist())):
    per_cu[kk].append(sd)
load = collections.Counter()
for kk, lst in per_cu.items():
    gj_per_simd = [0, 0, 0, 0]
    for sd in lst:
        gj_per_simd[sd[0]] += 1; gj_per_simd[sd[1]] += 1
    load[tuple(sorted(gj_per_simd))] += 1
print("CUs: %d; GJ waves per SIMD (sorted) histogram:" % len(per_cu), load.most_common(8))
piv = buf.reshape(len(ids), ncm * ncm)[full][:, 48:50]
npiv = 19 * 39  # 5x5-cell patch, n_sub 8: 39 interior lines of 39 dofs, 19 forward steps per chain (the meeting line is not counted)
print("GJ wave, shader cycles per pivot (sums over the forward sweep / %d pivots): write -> barrier -> read -> reciprocal -> scaled row %.0f, rank-1 update and replacements %.0f" % (
    npiv, piv[:, 0].mean() / npiv, piv[:, 1].mean() / npiv))
