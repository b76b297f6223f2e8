"""Point-sharded MSM across the GPUs of one node.

The reference is single-device (device index 0 everywhere: packages/backend/libs/src/utils/mod.rs:88-110), so
this layer is new.  The MSM sum is associative and commutative: every rank runs the full single-GPU
Pippenger over its contiguous range of points and the only exchange is ONE all_gather of world_size
144-byte partial results (RCCL over xGMI when the backend is "nccl"); every rank then adds the partials
with a world_size-point MSM with unit scalars, so the combine also runs through the HIP path.
RCCL has no reduce operator for group elements, hence all_gather + local add rather than all_reduce.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous [lo, hi) of rank's points; sizes differ by at most one"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_partials(dist, part144, device="cpu"):
    """all_gather of one 144-byte projective point per rank -> (world, 144) uint8 array on the host"""
    import torch
    mine = torch.from_numpy(np.ascontiguousarray(part144)).to(device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu().numpy()


def combine_partials(tkmk, partials):
    """sum of the gathered partial results through the GPU MSM (unit scalars)"""
    world = partials.shape[0]
    ones = np.zeros(32 * world, np.uint8)
    ones[0::32] = 1
    aff = np.concatenate([tkmk.projective_to_affine_bytes(np.ascontiguousarray(p)) for p in partials])
    return tkmk.msm(ones, aff)


def msm_sharded(tkmk, dist, scalars_shard, bases_shard, device="cuda"):
    """this rank's shard -> the full MSM result (144-byte canonical projective) on every rank"""
    part = tkmk.msm(scalars_shard, bases_shard)
    if dist is None or dist.get_world_size() == 1:
        return part
    return combine_partials(tkmk, gather_partials(dist, part, device))
