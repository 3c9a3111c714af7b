# Exploration (round 2): selection-stage decisions GPU vs oracle at oversampling 3, and the big
# register tiles (n_sub = 8..16 with oversampling 3) through whatever kernel the dispatch picks.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, slod_amd, slod_oracle as so

def sample_ids(g, per_shape=1):
    shapes = {}
    for pid in range(g.num_patches):
        i = g.patch_layout(pid)
        shapes.setdefault((i.mx, i.my, tuple(i.side_domain)), []).append(pid)
    return np.array(sorted(p for v in shapes.values() for p in v[:per_shape]), dtype=np.uint32)

def run(kw, dist, per_shape=1, cond=True, max_patches=None):
    cfg = so.make_cfg(**kw)
    g = slod_amd.Slod(**kw)
    d, lo, hi = {"D100": (0, 1.0, 100.0), "D1e4": (1, 1.0, 1.0e4)}[dist]
    field = so.fill_coefficient(20250614, d, lo, hi, g.NE)
    g.set_coefficient(0, field)
    ids = sample_ids(g, per_shape)
    if max_patches:
        ids = ids[:max_patches]
    sizes = [g.patch_layout(int(p)).n_fine for p in ids]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    total = int(np.sum(sizes))
    plan = g.plan(ids, offs)
    dev = torch.device("cuda", 0)
    b = torch.zeros(total, dtype=torch.float64, device=dev); q = torch.zeros_like(b)
    t0 = time.time()
    plan.execute(b.data_ptr(), q.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t1 = time.time() - t0
    try:
        plan.status(); st = "ok"
    except Exception as e:
        st = str(e)
    plan.execute(b.data_ptr(), q.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("== %s %s: %d patches, first execute %.3f s, status %s, kernel ms %s" % (kw, dist, len(ids), t1, st, plan.kernel_ms()), flush=True)
    dg = plan.diagnostics()
    hb = b.cpu().numpy()
    for k, pid in enumerate(ids):
        p = so.patch_info(cfg, int(pid))
        phi0, _, d0 = so.patch_basis(cfg, [field], int(pid))
        got = hb[int(offs[k]):int(offs[k]) + p.n_f]
        err = np.abs(got - phi0.ravel()).max()
        line = "%4d %dx%d %s nc %d nb %d | oracle cut %d drop %d dinf %.3e smax %.2e smin %.2e | gpu path %d cut %d drop %d dinf %.3e sw %d | err %.2e" % (
            pid, p.mx, p.my, list(p.side_domain), p.n_c, p.n_b, d0.n_cut[0], d0.n_dropped[0], d0.dinf[0], d0.sigma_max[0], d0.sigma_min[0],
            dg[k].path, dg[k].n_cut, dg[k].n_dropped, dg[k].dinf, dg[k].sweeps, err)
        if cond:
            sp, stable = so.selection_conditioning(cfg, [field], int(pid))
            line += " | cond spread %.2e stable %s" % (sp, stable)
        print(line, flush=True)

which = sys.argv[1:] or ["l3", "tiles", "c3"]
if "l3" in which:
    run(dict(nref=4, n_sub=4, oversampling=3, spacedim=1, stabilize=1), "D100", per_shape=2)
    run(dict(nref=4, n_sub=4, oversampling=3, spacedim=1, stabilize=1), "D1e4", per_shape=2)
    run(dict(nref=4, n_sub=4, oversampling=2, spacedim=1, stabilize=1), "D1e4", per_shape=1)
if "tiles" in which:
    for n in (8, 10, 12):
        run(dict(nref=3, n_sub=n, oversampling=3, spacedim=1, stabilize=1), "D100", cond=False, max_patches=6)
if "c3" in which:
    run(dict(nref=3, n_sub=16, oversampling=3, spacedim=1, stabilize=1), "D100", cond=False, max_patches=8)
