"""Worker of tests/test_distributed.py: one rank of a world_size-N gloo job on the CPU.
The HIP product path cannot run here, so the oracle stands in for the per-rank compute (test
infrastructure only); what is under test is the sharding + slab layout + all-gather logic that
bench.py uses on the GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-slod_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import slod_oracle as so                  # noqa: E402
from slod_amd import distributed as sd    # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(nref=2, n_sub=2, oversampling=1, spacedim=1, stabilize=1)
    n_problems = int(os.environ.get("SLOD_TEST_PROBLEMS", "1"))
    cfg = so.make_cfg(**kw)
    NP = so.num_patches(cfg)
    total = NP * n_problems
    stride = (2 * 3 + 1) ** 2                       # full patch n_fine, spacedim 1
    begin, end = sd.shard(total, world, rank)
    fields = [so.fill_coefficient(7 + pb, 0, 1.0, 100.0, 8) for pb in range(n_problems)]
    local_b = sd.allocate_slab(total, world, stride, "cpu")
    local_p = sd.allocate_slab(total, world, stride, "cpu")
    for k, gid in enumerate(range(begin, end)):
        phi, psi, _ = so.patch_basis(cfg, [fields[gid // NP]], gid % NP)
        local_b[k * stride:k * stride + phi.size] = torch.from_numpy(phi.ravel())
        local_p[k * stride:k * stride + psi.size] = torch.from_numpy(psi.ravel())
    gb = sd.all_gather_slabs(local_b, world)
    gp = sd.all_gather_slabs(local_p, world)
    ok = True
    if rank == 0:
        for gid in range(total):
            phi, psi, _ = so.patch_basis(cfg, [fields[gid // NP]], gid % NP)
            off = sd.global_offset(gid, total, world, stride)
            ok &= bool(np.array_equal(gb[off:off + phi.size].numpy(), phi.ravel()))
            ok &= bool(np.array_equal(gp[off:off + psi.size].numpy(), psi.ravel()))
        covered = sum(sd.shard(total, world, r)[1] - sd.shard(total, world, r)[0] for r in range(world))
        ok &= covered == total
        print("DIST_OK" if ok else "DIST_FAIL", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
