"""Row-block GEMV and chunked dot across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests). The reference has no multi-GPU code at all
(single device, stream 0: `/root/reference/cuda/dot_kernels.cuh:33`), so this is
new surface designed for the MI355X node:

  * GEMV shards by contiguous row blocks: rank g computes
    res[lo:hi] = alpha * A[lo:hi, :] * x + beta * res[lo:hi] with a replica of
    x. No reduction is needed; `gather=True` adds ONE all_gather of the result
    slices (8 x 16 KB at n = 32768: latency-bound, not bandwidth-bound).
  * dot shards by contiguous chunks: each rank reduces its chunk to a device
    scalar of the ARITHMETIC type, then ONE all_reduce(SUM) of that single
    element; the cast to the result type happens after the all-reduce, so the
    exchange never loses precision to the storage type. A 4/8-byte message is
    pure latency on xGMI; ring bandwidth is irrelevant here.

The local compute is injected (`local_gemv` / `local_dot` callables) so that
the same sharding logic runs on the HIP kernels (product) and, in the gloo
tests, on the CPU oracle.
"""


def shard_bounds(total, world_size, rank):
    """Contiguous block [lo, hi) of `total` items owned by `rank`; blocks differ
    by at most one item and cover [0, total) exactly."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad world_size / rank")
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def sharded_dot(local_dot, n, dist=None, group=None):
    """`local_dot(lo, hi)` must return a 1-element tensor of the arithmetic type
    holding sum_{lo <= i < hi} x_i * y_i on the communicator's device. Returns
    that tensor after the all-reduce (every rank holds the full dot)."""
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    lo, hi = shard_bounds(n, world, rank)
    partial = local_dot(lo, hi)
    if world > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def sharded_gemv(local_gemv, rows, dist=None, group=None, gather=False, res_full=None):
    """`local_gemv(lo, hi)` computes rows [lo, hi) in place in this rank's slice
    and returns that slice (a tensor of hi - lo elements). With gather=True the
    slices are all-gathered into `res_full` (rows elements, pre-allocated); rows
    must then divide evenly."""
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    lo, hi = shard_bounds(rows, world, rank)
    piece = local_gemv(lo, hi)
    if gather and world > 1:
        if rows % world != 0:
            raise ValueError("gather needs rows divisible by the number of ranks")
        dist.all_gather_into_tensor(res_full, piece, group=group)
        return res_full
    if gather:
        res_full.copy_(piece)
        return res_full
    return piece
