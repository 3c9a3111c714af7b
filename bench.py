#!/usr/bin/env python3
"""bench.py -- SLOD basis-construction throughput on MI355X (one JSON line on rank 0).

Metric (BASELINE.json / SURVEY.md section 8d): SLOD patches/s (basis built) for the
north-star configuration C2 = 2-D Poisson, H=1/32, n_sub=8, oversampling 2, SLOD,
random log-uniform coefficient of contrast 1e4 (D1e4, splitmix64 seed 20250614), fp64.

A "step" = one pass of the hot path over the rank's whole patch list: stencil assembly,
constrained 25-RHS patch solve, boundary trace + SVD selection, normalisation and
premultiplication for every patch (reference source/LOD.cc:345-767), inputs (the
coefficient field) and outputs (phi, psi) resident in HBM.

N GPUs (weak scaling): the workload is an ensemble of N independent coefficient
realisations of C2 (N*1024 patches); the global patch list is split in contiguous blocks
exactly like Utilities::MPI::create_evenly_distributed_partitioning (LOD.cc:116-118), one
block per rank, no data-path collective inside the timed basis build.  After the timed
region every rank's (phi,psi) slab is all-gathered over RCCL (north_star's exchange step);
that time is reported separately as `allgather_ms` (SURVEY 8d excludes it from the metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))

SEED = 20250614
C2 = dict(nref=5, n_sub=8, oversampling=2, spacedim=1, stabilize=1)
# the other BASELINE.json configurations (extra bench lines, never the headline):
#   C3  2-D Poisson H=1/128, n_sub=16, oversampling 3: 16384 patches SHARDED over the ranks (strong scaling)
#   C4  2-D elasticity H=1/32, n_sub=8, oversampling 2 (ensemble per rank like C2)
CONFIGS = {
    "C2": (C2, "weak", "C2: 2D Poisson SLOD, H=1/32, n_sub=8, oversampling=2"),
    "C3": (dict(nref=7, n_sub=16, oversampling=3, spacedim=1, stabilize=1), "strong",
           "C3: 2D Poisson SLOD, H=1/128, n_sub=16, oversampling=3"),
    "C4": (dict(nref=5, n_sub=8, oversampling=2, spacedim=2, stabilize=1), "weak",
           "C4: 2D linear elasticity SLOD, H=1/32, n_sub=8, oversampling=2"),
}
PEAK_FP64_TFLOPS = 78.6   # MI355X dense fp64 (vector = matrix): half the 157.3 TF fp32 vector peak
PEAK_HBM_GBPS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md


def canonical_counts(slod, gids):
    """Algorithmic flops of the patch solve (banded Cholesky in patch-lexicographic order,
    SURVEY 8d: N_I(b^2+3b) + 4 N_c N_I b) and algorithmic bytes 8*(4*E + 2*s*N_f)."""
    s, n = slod.spacedim, slod.cfg.n_subdivisions
    flops = 0.0
    nbytes = 0.0
    cache = {}
    for g in gids:
        pid = int(g) % slod.num_patches
        info = slod.patch_layout(pid)
        key = (info.mx, info.my)
        if key not in cache:
            b = s * (n * min(info.mx, info.my) - 1) + s - 1
            ni, nc = info.n_internal, info.n_coarse
            e = info.nx * info.ny
            cache[key] = (ni * (b * b + 3 * b) + 4.0 * nc * ni * b, 8.0 * (4 * e * s + 2 * s * info.n_fine))
        flops += cache[key][0]
        nbytes += cache[key][1]
    return flops, nbytes


def cpu_baseline(cfg_kw, fields, n_full, every=1):
    """The C oracle (kind 'port': our CPU restatement, the reference cannot be built here)
    timed on this box's host cores on a bounded sample of the same workload (every `every`-th
    patch of the configuration)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import slod_oracle as so
    cfg = so.make_cfg(**cfg_kw)
    s = cfg.spacedim
    ids = np.arange(0, n_full, every, dtype=np.int32)
    sizes = np.array([s * so.patch_info(cfg, int(p)).n_f for p in ids], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    total = int(sizes.sum())
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # a container CPU quota (cgroup v2) is the real core budget
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    # 1 thread: the reference's execution model per rank (MPI_InitFinalize(...,1)); 128 patches
    sub = ids[::8]
    t0 = time.perf_counter()
    so.basis_many(cfg, fields, sub, offs[::8], total, nthreads=1)
    t1 = time.perf_counter() - t0
    # all cores, OpenMP over patches: the reference's scale-out model (locally_owned_patches)
    reps = 0
    t0 = time.perf_counter()
    while True:
        so.basis_many(cfg, fields, ids, offs, total, nthreads=cores)
        reps += 1
        tall = time.perf_counter() - t0
        if tall > 6.0 or reps >= 400:
            break
    return {"value": reps * len(ids) / tall, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": "%d patches (every %d-th of the configuration) x %d passes, OpenMP over patches "
                      "(%.1f s wall); 1 thread on every 8th of those: %.1f patches/s"
                      % (len(ids), every, reps, tall, len(sub) / t1),
            "value_1thread": len(sub) / t1}


def lod_system_leg(slod, torch, dev, basis, premult, stride):
    """Device times of the SURVEY 8(f) rows on the bench configuration: global LOD matrix and right-hand
    side from the (phi, psi) slab just built (LOD.cc:860-973,982), coarse CG solve (:990-998),
    reconstruction u_h = C u_H (:1251) and the fine FEM reference solve (:1004-1094), f = 1.  Achieved
    GB/s against the algorithmic bytes each kernel must move (stated per row)."""
    s, NP, NE = slod.spacedim, slod.num_patches, slod.NE
    cap = slod.lod_row_capacity()
    rows = np.arange(NP, dtype=np.uint32)
    nfine = (NE + 1) * (NE + 1) * s

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    vals = torch.zeros(NP * cap * s * s, dtype=torch.float64, device=dev)
    cols = torch.zeros(NP * cap, dtype=torch.int32, device=dev)
    frhs = torch.zeros(nfine, dtype=torch.float64, device=dev)
    rhs = torch.zeros(NP * s, dtype=torch.float64, device=dev)
    u = torch.zeros(NP * s, dtype=torch.float64, device=dev)
    fine = torch.zeros(nfine, dtype=torch.float64, device=dev)
    ufem = torch.zeros(nfine, dtype=torch.float64, device=dev)
    slab_bytes = 8.0 * NP * stride
    t_mat, _ = timed(lambda: slod.lod_matrix(rows, basis.data_ptr(), premult.data_ptr(), stride, vals.data_ptr(), cols.data_ptr()))
    slod.fem_rhs(None, frhs.data_ptr())
    t_rhs, _ = timed(lambda: slod.lod_rhs(rows, basis.data_ptr(), stride, frhs.data_ptr(), rhs.data_ptr()))
    t_sol, (it_lod, res_lod) = timed(lambda: slod.lod_solve(vals.data_ptr(), cols.data_ptr(), rhs.data_ptr(), u.data_ptr(), 1e-10, 20000), reps=1)
    t_rec, _ = timed(lambda: slod.lod_reconstruct(basis.data_ptr(), stride, u.data_ptr(), fine.data_ptr()))
    t_fem, (it_fem, res_fem) = timed(lambda: slod.fem_solve(frhs.data_ptr(), ufem.data_ptr(), 1e-10, 100000), reps=1)
    err = float((fine - ufem).norm() / ufem.norm())
    gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9
    return {
        "lod_matrix_ms": t_mat, "lod_matrix_GBps": gbps(2 * slab_bytes + 8.0 * NP * cap * (s * s + 0.5), t_mat),
        "lod_rhs_ms": t_rhs, "lod_rhs_GBps": gbps(slab_bytes + 8.0 * nfine, t_rhs),
        "lod_solve": {"iters": it_lod, "ms": t_sol, "rel_residual": res_lod, "preconditioner": "Jacobi",
                      "GBps": gbps(it_lod * 8.0 * NP * cap * (s * s + 0.5), t_sol)},
        "reconstruct_ms": t_rec, "reconstruct_GBps": gbps(slab_bytes + 8.0 * nfine, t_rec),
        "fem_solve": {"iters": it_fem, "ms": t_fem, "rel_residual": res_fem,
                      "preconditioner": os.environ.get("SLOD_FEM_PRECOND", "multigrid V(2,2), Galerkin, damped Jacobi" if s == 1 else "Jacobi"),
                      "unknowns": nfine, "GBps": gbps(it_fem * 8.0 * nfine * (9 * s * s + 6), t_fem)},
        "rel_l2_lod_vs_fem": err,
        "bytes_model": "lod_matrix: (phi,psi) slab once + the block rows written; lod_rhs/reconstruct: phi slab + fine vector; "
                       "CG: per iteration the block rows (values + column ids) resp. the 9-point stencil planes + 6 vectors",
        "hbm_peak_GBps": PEAK_HBM_GBPS,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two-stream measurement")
    ap.add_argument("--no-lod-system", action="store_true", help="skip the timing of the consumers of (phi, psi)")
    ap.add_argument("--dist", default="D1e4", choices=["D100", "D1e4"])
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (C2 = the headline metric's; C3, C4: extra lines)")
    args = ap.parse_args()
    CFG, scaling, cfg_label = CONFIGS[args.config]

    import torch
    import slod_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- workload.  weak: ensemble of `world` realisations of the configuration; strong (C3):
    #      ONE realisation whose patches are sharded.  Either way: this rank's contiguous block
    n_prob = world if scaling == "weak" else 1
    slod = slod_amd.Slod(n_problems=n_prob, device=local_rank, **CFG)
    NP = slod.num_patches
    total = NP * n_prob
    begin, end = slod_amd.partition(total, world, rank)     # LOD.cc:116-118
    gids = np.arange(begin, end, dtype=np.uint32)
    from slod_amd.synthetic import fill_coefficient
    lo, hi = (1.0, 100.0) if args.dist == "D100" else (1.0, 1.0e4)
    probs = sorted(set(int(g) // NP for g in gids))
    fields = {}
    for pb in probs:
        fields[pb] = [fill_coefficient(SEED + 1000 * pb + f, args.dist, slod.NE) for f in range(slod.spacedim)]
        for f in range(slod.spacedim):                            # elasticity: lambda = seed, mu = seed + 1
            t = torch.from_numpy(fields[pb][f]).to(dev)           # resident in HBM before timing
            slod.set_coefficient_device(f, t.data_ptr(), t.numel(), problem=pb)
    from slod_amd import distributed as sd
    plan = slod.plan(gids)                                        # uniform stride -> all-gather slabs
    n_local = len(gids)
    basis = sd.allocate_slab(total, world, plan.stride, dev)      # padded slab (ragged tail)
    premult = sd.allocate_slab(total, world, plan.stride, dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.execute(basis.data_ptr(), premult.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    plan.status()

    plan.profile(args.steps)          # HIP events around every launch of the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    plan.status()
    # mean per-kernel device time over the timed region (events on the launch stream)
    ks = np.array(plan.kernel_ms())

    # outside the metric: the same K steps double-buffered over two plans / two HIP streams.  The
    # passes are independent, so the next pass fills the CUs that the last slow patches (SVD
    # fallback) of the previous one leave idle; reported next to the serial headline number.
    pipelined = None
    if world == 1 and not args.no_pipeline:
        ref_b, ref_q = basis.clone(), premult.clone()
        try:
            plan.set_overlap(2)           # library-level: second workspace + two internal streams
        except slod_amd.SlodError:        # plans in several workspace chunks (C3) de-phase by themselves
            plan = plan
        else:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ep = time.perf_counter() - tp
            plan.status()
            assert torch.equal(ref_b, basis) and torch.equal(ref_q, premult)
            pipelined = {"api": "slod_plan_set_overlap(plan, 2): consecutive slod_plan_execute calls alternate between two "
                                "workspaces / internal streams", "value": n_local * args.steps / ep,
                         "ms_per_step": ep / args.steps * 1e3}
            plan.set_overlap(1)
        del ref_b, ref_q

    # outside the metric: the consumers of (phi, psi) at this configuration's scale:
    # A_LOD = C^T (A C), C^T f, coarse solve, reconstruction, fine FEM reference solve
    lod_system = None
    if world == 1 and not args.no_lod_system and n_prob == 1:
        lod_system = lod_system_leg(slod, torch, dev, basis, premult, plan.stride)

    # the exchange step (outside the metric): RCCL all-gather of the (phi,psi) slabs
    allgather_ms = None
    if world > 1:
        for _ in range(2):
            gb = sd.all_gather_slabs(basis, world)
            gp = sd.all_gather_slabs(premult, world)
        barrier()
        ta = time.perf_counter()
        gb = sd.all_gather_slabs(basis, world)
        gp = sd.all_gather_slabs(premult, world)
        barrier()
        allgather_ms = (time.perf_counter() - ta) * 1e3
        # every rank's block must land where a single-GPU run would put it
        mine = gb[rank * basis.numel():(rank + 1) * basis.numel()]
        assert torch.equal(mine, basis)
        tmax = torch.tensor([elapsed, allgather_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, allgather_ms = float(tmax[0]), float(tmax[1])

    # The same exchange through the C-ABI (what a C++ host calls: slod_comm_* over RCCL,
    # slod_plan_execute_allgather computes the rank's block in pieces and broadcasts every finished
    # piece on a second stream while the next one computes).  Outside the metric.  The gathered slabs
    # must equal the torch.distributed result bit for bit.  A watchdog prints the headline line and
    # leaves if this leg does not come back (it has never run on more than one GPU before the
    # driver's scaling run).
    abi_exchange = None
    headline = {}
    if world > 1:
        import threading

        def bail():
            if rank == 0 and headline:
                headline["abi_exchange"] = {"error": "slod_plan_execute_allgather did not return within 60 s"}
                print(json.dumps(headline), flush=True)
            os._exit(0 if headline or rank != 0 else 1)
        watchdog = threading.Timer(60.0, bail)
        watchdog.daemon = True
    else:
        watchdog = None

    def run_abi_exchange():
        ident = [slod_amd.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        comm = slod_amd.Comm(ident[0], world, rank, local_rank)
        ppr = sd.slab_patches(total, world)
        gb2 = torch.zeros(world * basis.numel(), dtype=torch.float64, device=dev)
        gp2 = torch.zeros_like(gb2)
        cs, xs = torch.cuda.Stream(), torch.cuda.Stream()
        res = {}
        for n_pieces in (1, 4):
            for rep in range(2):
                barrier()
                ta = time.perf_counter()
                plan.execute_allgather(comm, gb2.data_ptr(), gp2.data_ptr(), ppr, n_pieces, cs.cuda_stream, xs.cuda_stream)
                barrier()
                te = (time.perf_counter() - ta) * 1e3
            plan.status()
            ok = bool(torch.equal(gb2, gb)) and bool(torch.equal(gp2, gp))
            tt = torch.tensor([te, 0.0 if ok else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            res["pieces_%d" % n_pieces] = {"step_plus_exchange_ms": float(tt[0]), "equals_torch_allgather": float(tt[1]) == 0.0}
        comm.close()
        return res

    if rank == 0:
        flops, nbytes = canonical_counts(slod, gids)
        ms_step = elapsed / args.steps * 1e3
        solve_s = ks[1] * 1e-3
        achieved_tf = flops / solve_s / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == "C2":
            try:
                traffic = json.load(open(tpath)).get("k_solve_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "SLOD patches/sec (basis built), 2D Poisson H=1/32 n_sub=8 oversampling 2" if args.config == "C2"
                      else "SLOD patches/sec (basis built), " + cfg_label,
            "value": total * args.steps / elapsed,
            "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg_label + ", %s coefficient (contrast %g), %d patches on rank 0 (%s, contiguous "
                                   "patch blocks)" % (args.dist, hi / lo, n_local,
                                                      "ensemble of %d realisations" % world if scaling == "weak"
                                                      else "%d patches sharded over %d ranks" % (total, world)),
                       "patches_per_gpu": n_local, "parallelism": "patch-sharded x%d" % world},
            # the kernel's roof is the fp64 rate (on MI355X the vector and the matrix pipe have the
            # same 78.6 TFLOP/s): "mfma" names that peak; what actually limits the kernel today is
            # the instruction issue / dependent-chain latency of its Gauss-Jordan waves
            "roofline": {"bound": "mfma",
                         "limiter": "forward sweep: vector issue (two Gauss-Jordan waves + two helpers per SIMD, 57 VALU per pivot) and the CU's LDS unit on the pivot-row broadcast; backward sweep: workspace loads of V/Z per line (DESIGN section 6, measured on the device)",
                         "kernel": "k_solve_%s" % os.environ.get("SLOD_SOLVE", "tw") + (" (stencil assembly and selection stage fused in)" if ks[2] < 0.05 * ks[1] and ks[0] < 0.05 * ks[1] else (" (selection stage fused in)" if ks[2] < 0.05 * ks[1] else "")),
                         "achieved": achieved_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "algorithmic_flops_per_launch": flops,
                         "kernel_ms": {"assemble": ks[0], "solve": ks[1], "select": ks[2]},
                         "hbm_algorithmic_bytes_per_step": nbytes,
                         "hbm_achieved_GBps": nbytes / (ms_step * 1e-3) / 1e9,
                         "hbm_frac": nbytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBPS},
            "allgather_ms": allgather_ms,
            "pipelined": pipelined,
            "lod_system": lod_system,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(CFG, fields[probs[0]], NP, every={"C2": 1, "C3": 256, "C4": 8}[args.config])
        headline.update(out)
    if world > 1:
        dist.barrier()            # rank 0 comes out of its CPU-baseline leg here
        watchdog.start()
        try:
            abi_exchange = run_abi_exchange()
        except Exception as e:    # noqa: BLE001 -- reported in the line, never fatal for the metric
            abi_exchange = {"error": "%s: %s" % (type(e).__name__, e)}
        watchdog.cancel()
    if rank == 0:
        out["abi_exchange"] = abi_exchange
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
