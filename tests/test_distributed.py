"""N > 1 path on CPU: 2 ranks over gloo run the same sharding logic the GPU bench
uses (accblas.multi_gpu), with the oracle as the local compute. Checks that the
row-block GEMV needs no reduction and equals the single-process result bit for
bit, and that the chunked dot + one all-reduce matches the full dot."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as oracle
from accblas import multi_gpu
from oracle_lib import F32, F64


def test_shard_bounds_cover_exactly():
    for total in (0, 1, 7, 64, 1000, 32768, 2 ** 28):
        for world in (1, 2, 3, 4, 8):
            bounds = [multi_gpu.shard_bounds(total, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == total
            for (lo, hi), (lo2, _) in zip(bounds, bounds[1:]):
                assert hi == lo2 and hi >= lo
            sizes = [hi - lo for lo, hi in bounds]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        multi_gpu.shard_bounds(10, 2, 2)
    # aligned starts (16 bytes of the storage type / 64-row GEMV blocks)
    for total, world, align in [(100_001, 3, 8), (32768, 8, 64), (1 << 28, 8, 8), (100, 8, 64)]:
        bounds = [multi_gpu.shard_bounds(total, world, r, align) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
        assert all(lo % align == 0 for lo, _ in bounds)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)  # same data on every rank
        rows, cols = 64, 300
        A = oracle.convert(rng.uniform(-1, 1, rows * cols), F32)
        x = oracle.convert(rng.uniform(-1, 1, cols), F32)
        res0 = oracle.convert(rng.uniform(-1, 1, rows), F32)
        res = res0.copy()

        def local_gemv(lo, hi):
            oracle.accessor_gemv_rows(F64, lo, hi, cols, 1.0, A, cols, x, 1, 1.0, res, 1)
            return torch.from_numpy(res[lo:hi])

        full = torch.empty(rows, dtype=torch.float32)
        multi_gpu.sharded_gemv(local_gemv, rows, dist=dist, gather=True, res_full=full)
        single = res0.copy()
        oracle.accessor_gemv_rows(F64, 0, rows, cols, 1.0, A, cols, x, 1, 1.0, single, 1)
        assert np.array_equal(full.numpy(), single)

        # the bench.py code path for N > 1 (strong scaling: the fixed problem
        # split into 64-row-aligned blocks, one all-gather of the slices): bits
        # equal the single-rank product
        import bench
        rows_b = 128
        A_b = oracle.convert(rng.uniform(-1, 1, rows_b * cols), F32)
        res0_b = oracle.convert(rng.uniform(-1, 1, rows_b), F32)
        res_b = res0_b.copy()

        def local_gemv_b(lo, hi):
            oracle.accessor_gemv_rows(F64, lo, hi, cols, 1.0, A_b, cols, x, 1, 1.0, res_b, 1)
            return torch.from_numpy(res_b[lo:hi])

        full_b = torch.empty(rows_b, dtype=torch.float32)
        lo, hi, seconds = bench.strong_gemv(local_gemv_b, rows_b, dist, steps=1, warmup=0,
                                            barrier=dist.barrier, gather_into=full_b)
        assert (lo, hi) == (64 * rank, 64 * rank + 64) and seconds > 0
        single_b = res0_b.copy()
        oracle.accessor_gemv_rows(F64, 0, rows_b, cols, 1.0, A_b, cols, x, 1, 1.0, single_b, 1)
        assert np.array_equal(full_b.numpy(), single_b)
        # without the gather there is no collective: every rank holds its slice only
        res_b[:] = res0_b
        lo, hi, _ = bench.strong_gemv(local_gemv_b, rows_b, dist, steps=1, warmup=0,
                                      barrier=dist.barrier)
        assert np.array_equal(res_b[lo:hi], single_b[lo:hi])
        other = slice(0, 64) if rank == 1 else slice(64, 128)
        assert np.array_equal(res_b[other], res0_b[other])

        n = 100_001
        xv = oracle.convert(rng.uniform(-1, 1, n), F32)
        yv = oracle.convert(rng.uniform(-1, 1, n), F32)

        def local_dot(lo, hi):
            return torch.tensor([oracle.accessor_dot_chunk(F64, lo, hi, xv, 1, yv, 1)],
                                dtype=torch.float64)

        total = multi_gpu.sharded_dot(local_dot, n, dist=dist, align=4)
        assert multi_gpu.shard_bounds(n, world, 1, 4)[0] % 4 == 0
        want, scale = oracle.exact_dot(F64, n, xv, 1, yv, 1)
        assert abs(total.item() - want) <= 64 * 2.0 ** -53 * scale
        # every rank holds the same reduced value
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, total)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_gemv_and_dot(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_single_process_path_needs_no_communicator():
    got = multi_gpu.sharded_dot(lambda lo, hi: torch.tensor([float(hi - lo)]), 10)
    assert got.item() == 10.0
