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


def shard_bounds(total, world_size, rank, align=1):
    """Contiguous block [lo, hi) of `total` items owned by `rank`; the blocks
    cover [0, total) exactly and, with align = 1, differ by at most one item.
    With align > 1 every interior block edge is rounded down to a multiple of
    `align` items (the same rule as `accblas_shard_bounds` of
    include/accblas_comm.h): dot chunks use align = 16 bytes / sizeof(storage)
    so that every chunk keeps the 16-byte vector loads (an unaligned start
    silently drops to the scalar path), GEMV row blocks use align = 64 rows so
    that a row's summation order is the same alone as inside the full
    product."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad world_size / rank")
    if align < 1:
        raise ValueError("align must be >= 1")
    base, extra = divmod(total, world_size)

    def edge(r):
        if r >= world_size:
            return total
        e = r * base + min(r, extra)
        return e - e % align

    return edge(rank), edge(rank + 1)


def sharded_dot(local_dot, n, dist=None, group=None, align=1):
    """`local_dot(lo, hi)` must return a 1-element tensor of the arithmetic type
    holding sum_{lo <= i < hi} x_i * y_i on the communicator's device. Returns
    that tensor after the all-reduce (every rank holds the full dot). `align`:
    see `shard_bounds` (16 bytes / sizeof(storage) keeps the vector path)."""
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    lo, hi = shard_bounds(n, world, rank, align)
    partial = local_dot(lo, hi)
    if world > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def sharded_gemv(local_gemv, rows, dist=None, group=None, gather=False, res_full=None,
                 align=1):
    """`local_gemv(lo, hi)` computes rows [lo, hi) in place in this rank's slice
    and returns that slice (a tensor of hi - lo elements). With gather=True the
    slices are all-gathered into `res_full` (rows elements, pre-allocated); the
    row blocks must then all have the same size."""
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    lo, hi = shard_bounds(rows, world, rank, align)
    piece = local_gemv(lo, hi)
    if gather and world > 1:
        if rows % (world * align) != 0:
            raise ValueError("gather needs rows divisible by the number of ranks")
        dist.all_gather_into_tensor(res_full, piece, group=group)
        return res_full
    if gather:
        res_full.copy_(piece)
        return res_full
    return piece
