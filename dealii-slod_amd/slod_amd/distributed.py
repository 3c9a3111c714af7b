"""Multi-GPU plumbing: contiguous patch sharding + RCCL all-gather of the basis slabs.

The reference shards patches with Utilities::MPI::create_evenly_distributed_partitioning
(source/LOD.cc:116-118) and never communicates the basis (its MPI path is unfinished,
LOD.cc:225-229).  Here every rank builds its contiguous block of patches with no data-path
collective, then ONE all-gather per array (backend "nccl" = RCCL over xGMI on the GPUs, "gloo"
in the CPU tests) replicates (phi, psi) on all ranks.  Slabs use the plan's uniform per-patch
stride and are padded to ceil(total/world) patches so that all ranks contribute equal sizes.
"""
import torch
import torch.distributed as dist

from . import partition


def shard(total_patches, world, rank):
    """[begin, end) of the rank's contiguous block of global patch ids."""
    return partition(total_patches, world, rank)


def slab_patches(total_patches, world):
    return (total_patches + world - 1) // world


def allocate_slab(total_patches, world, stride, device):
    n = slab_patches(total_patches, world) * stride
    return torch.zeros(n, dtype=torch.float64, device=device)


def all_gather_slabs(local, world):
    """local: this rank's padded slab.  Returns the concatenation of all ranks' slabs."""
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if world == 1:
        out.copy_(local)
    else:
        dist.all_gather_into_tensor(out, local)
    return out


def global_offset(gid, total_patches, world, stride):
    """Offset (in doubles) of global patch `gid` inside the gathered array."""
    q, r = divmod(total_patches, world)
    # rank owning gid under the evenly distributed partitioning
    if gid < r * (q + 1):
        rank, local = divmod(gid, q + 1)
    else:
        rank, local = divmod(gid - r * (q + 1), q) if q else (world - 1, 0)
        rank += r
    return (rank * slab_patches(total_patches, world) + local) * stride
