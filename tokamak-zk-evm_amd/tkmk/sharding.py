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


# ---------------------------------------------------------------------------------------------------
# One large bivariate NTT sharded over the ranks: the only sub-path with a real exchange.
#   rank r holds the x-slab: rows [r*xs/G, (r+1)*xs/G) of the xs x ys matrix (element (ix,iy) at ix*ys + iy)
#   1. row transforms (length ys, coset_y) on the slab                                — local
#   2. ONE all_to_all: block (rows of r) x (columns of q) goes to rank q              — RCCL over xGMI (nccl backend)
#   3. column transforms (length xs, coset_x) on the assembled xs x (ys/G) y-slab     — local
# Output = this rank's y-slab (all ix, columns [r*ys/G, (r+1)*ys/G)), row-major xs x (ys/G).
# Independent NTT batches (u/v/w, the leaves of p_comb) shard by batch index with no communication at all.
# ---------------------------------------------------------------------------------------------------
class TkmkNttOps:
    """per-rank compute through the HIP library; buffers are host numpy arrays here (the exchange is the point)"""

    def __init__(self, tkmk):
        self.t = tkmk

    def ntt_rows(self, buf, n, batch, coset, inverse):
        return self.t.ntt(buf, n, batch=batch, inverse=inverse, coset_gen=coset)

    def ntt_cols(self, buf, n, batch, coset, inverse):
        return self.t.ntt(buf, n, batch=batch, columns_batch=True, inverse=inverse, coset_gen=coset)


def bintt_sharded(ops, dist, slab, x_size, y_size, inverse=False, coset_x=None, coset_y=None, device="cpu"):
    import torch
    G, r = dist.get_world_size(), dist.get_rank()
    if x_size % G or y_size % G:
        raise ValueError("x_size and y_size must be multiples of the world size")
    rows, cols = x_size // G, y_size // G
    assert slab.size == 32 * rows * y_size
    # 1. rows of this slab
    a = ops.ntt_rows(np.ascontiguousarray(slab), y_size, rows, coset_y, inverse)
    # 2. all_to_all: send[q] = block (rows x cols_q); recv[q] = block from rank q (its rows, my columns)
    m = a.reshape(rows, G, cols * 32)
    send = torch.from_numpy(np.ascontiguousarray(m.transpose(1, 0, 2))).to(device)      # [q][row][col bytes]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    yslab = recv.cpu().numpy().reshape(G * rows, cols * 32).reshape(-1)                  # rows of rank 0, then rank 1, ... = all ix
    # 3. columns of the y-slab: xs x cols, element (ix, j) at ix*cols + j
    return ops.ntt_cols(np.ascontiguousarray(yslab), x_size, cols, coset_x, inverse)


def shard_batch(n_batch, rank, world):
    """independent NTTs: contiguous batch range of this rank (no communication)"""
    return shard_range(n_batch, rank, world)


# ---------------------------------------------------------------------------------------------------
# Independent commits of one prover round distributed over the ranks: the rounds are
# sequential (Fiat-Shamir), but inside a round the 6 (prove0) / 9 (prove4) encode_poly MSMs are independent.
# Job j runs on rank j mod G (through the pipelined multi-MSM of that GPU); ONE all_gather of ceil(J/G) results
# per rank returns every commitment to every rank, in job order.
# ---------------------------------------------------------------------------------------------------
def jobs_of_rank(n_jobs, rank, world):
    return list(range(rank, n_jobs, world))


def commits_sharded(msm_multi, dist, jobs, result_bytes=144, device="cuda"):
    """jobs: the same list on every rank (each rank only touches the operands of its own jobs);
    msm_multi(list of jobs) -> concatenated results of result_bytes each.  Returns (n_jobs, result_bytes) uint8."""
    import torch
    n = len(jobs)
    if dist is None or dist.get_world_size() == 1:
        return np.asarray(msm_multi(jobs)).reshape(n, result_bytes)
    G, r = dist.get_world_size(), dist.get_rank()
    mine = jobs_of_rank(n, r, G)
    slots = (n + G - 1) // G
    buf = np.zeros((slots, result_bytes), np.uint8)
    if mine:
        buf[:len(mine)] = np.asarray(msm_multi([jobs[j] for j in mine])).reshape(len(mine), result_bytes)
    t = torch.from_numpy(buf).to(device)
    out = [torch.empty_like(t) for _ in range(G)]
    dist.all_gather(out, t)
    res = np.zeros((n, result_bytes), np.uint8)
    for q in range(G):
        got = out[q].cpu().numpy()
        for k, j in enumerate(jobs_of_rank(n, q, G)):
            res[j] = got[k]
    return res


def balanced_assignment(sizes, world):
    """jobs -> ranks by size: largest first onto the least loaded rank (ties to the lower rank), the same on every rank.
    A prover round mixes 4 M-point and few-hundred-point commits, so j mod G would leave ranks idle.  -> [rank of job j]"""
    load = [0] * world
    owner = [0] * len(sizes)
    for j in sorted(range(len(sizes)), key=lambda k: (-int(sizes[k]), k)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[j] = r
        load[r] += int(sizes[j])
    return owner


def commits_balanced(msm_multi, dist, jobs, sizes, result_bytes=144, device="cuda"):
    """commits_sharded with balanced_assignment(sizes): ONE all_gather of max-jobs-per-rank results; every rank returns all
    commitments in job order.  jobs[j] is only touched on its owner."""
    import torch
    n = len(jobs)
    if dist is None or dist.get_world_size() == 1 or n == 0:
        return np.asarray(msm_multi(jobs)).reshape(n, result_bytes)
    G, r = dist.get_world_size(), dist.get_rank()
    owner = balanced_assignment(sizes, G)
    per_rank = [[j for j in range(n) if owner[j] == q] for q in range(G)]
    slots = max(len(v) for v in per_rank)
    buf = np.zeros((slots, result_bytes), np.uint8)
    mine = per_rank[r]
    if mine:
        buf[:len(mine)] = np.asarray(msm_multi([jobs[j] for j in mine])).reshape(len(mine), result_bytes)
    t = torch.from_numpy(buf).to(device)
    out = [torch.empty_like(t) for _ in range(G)]
    dist.all_gather(out, t)
    res = np.zeros((n, result_bytes), np.uint8)
    for q in range(G):
        got = out[q].cpu().numpy()
        for k, j in enumerate(per_rank[q]):
            res[j] = got[k]
    return res
