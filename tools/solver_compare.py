# Kernel time of the patch-solve families on a given configuration (all patches of the grid).
# usage: SLOD_SOLVE=mf|tw python tools/solver_compare.py nref n_sub oversampling spacedim [reps]
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
import numpy as np, torch, slod_amd
from slod_amd.synthetic import fill_coefficient
nref, n_sub, l, s = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = slod_amd.Slod(device=0, nref=nref, n_sub=n_sub, oversampling=l, spacedim=s, stabilize=1)
dev = torch.device("cuda", 0)
for f in range(s):
    t = torch.from_numpy(fill_coefficient(20250614 + f, "D100", g.NE)).to(dev)
    g.set_coefficient_device(f, t.data_ptr(), t.numel())
ids = np.arange(g.num_patches, dtype=np.uint32)
plan = g.plan(ids)
basis = torch.zeros(len(ids) * plan.stride, dtype=torch.float64, device=dev); premult = torch.zeros_like(basis)
plan.execute(basis.data_ptr(), premult.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); plan.status()
plan.profile(reps)
t0 = time.perf_counter()
for _ in range(reps):
    plan.execute(basis.data_ptr(), premult.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("%s nref %d n_sub %d l %d s %d: %d patches, %.3f ms per pass (%.0f patches/s), kernel ms %s" % (
    os.environ.get("SLOD_SOLVE", "auto"), nref, n_sub, l, s, len(ids), dt * 1e3, len(ids) / dt, plan.kernel_ms()), flush=True)
