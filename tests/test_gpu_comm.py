"""C-level multi-GPU context (include/accblas_comm.h) on the one GPU of the test
box: a world of one rank exercises communicator creation (both ways), the
sharded entry points and the driver binary; with one rank the results must equal
the single-GPU entry points bit for bit. More ranks run on the 8-GPU node only
(`bin/multi_gpu_benchmark --gpus=8`, `bench.py --gpus 8`)."""
import os
import subprocess

import numpy as np
import pytest
import torch

import accblas
from accblas import comm

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def handle():
    h = accblas.Handle()
    yield h
    h.close()


@pytest.fixture(scope="module", params=["unique_id", "create_all"])
def world_of_one(request):
    torch.cuda.set_device(0)
    if request.param == "unique_id":
        c = comm.Comm.create(1, 0, comm.unique_id())
    else:
        (c,) = comm.Comm.create_all(1)
    assert (c.rank, c.world_size, c.device) == (0, 1, 0)
    yield c
    c.close()


def test_comm_dot_equals_single_gpu_dot(world_of_one, handle):
    gen = torch.Generator(device="cpu").manual_seed(11)
    n = 3_000_001
    x = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).to(torch.float16).to(DEV)
    y = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).to(torch.float16).to(DEV)
    lo, hi = world_of_one.shard(n, 8)
    assert (lo, hi) == (0, n)
    single = torch.zeros(1, dtype=torch.float32, device=DEV)
    accblas.acc_dot(handle, accblas.F32, accblas.F16, accblas.F32, n, x, 1, y, 1, single)
    for res_dtype, res_t in [(torch.float32, accblas.F32), (torch.float64, accblas.F64)]:
        res = torch.full((1,), 7.0, dtype=res_dtype, device=DEV)
        world_of_one.acc_dot(handle, accblas.F32, accblas.F16, res_t, n, x, 1, y, 1, res)
        torch.cuda.synchronize()
        assert res.item() == single.item()  # cast after the exchange: exact widening
    # an empty chunk contributes zero
    res = torch.full((1,), 7.0, dtype=torch.float32, device=DEV)
    world_of_one.acc_dot(handle, accblas.F32, accblas.F16, accblas.F32, 0, x, 1, y, 1, res)
    torch.cuda.synchronize()
    assert res.item() == 0.0


def test_comm_gemv_equals_single_gpu_gemv(world_of_one):
    gen = torch.Generator(device="cpu").manual_seed(12)
    rows, cols = 777, 2051
    A = (torch.rand(rows, cols, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
    x = (torch.rand(cols, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
    init = (torch.rand(rows, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
    want = init.clone()
    accblas.acc_gemv(accblas.F64, accblas.F32, rows, cols, 0.5, A, cols, x, 1, -1.0, want, 1)
    local = init.clone()
    full = torch.zeros(rows, dtype=torch.float32, device=DEV)
    world_of_one.acc_gemv(accblas.F64, accblas.F32, rows, cols, 0.5, A, cols, x, 1, -1.0,
                          local, 1, res_full=full)
    torch.cuda.synchronize()
    assert torch.equal(local, want) and torch.equal(full, want)
    # the gather needs a contiguous slice
    with pytest.raises(accblas.AccblasError) as info:
        world_of_one.acc_gemv(accblas.F64, accblas.F32, rows // 2, cols, 1.0, A, cols, x, 1,
                              0.0, local, 2, res_full=full)
    assert info.value.status == -2


def test_comm_rejects_unsupported_arithmetic(world_of_one, handle):
    x = torch.zeros(8, dtype=torch.float16, device=DEV)
    res = torch.zeros(1, dtype=torch.float32, device=DEV)
    with pytest.raises(accblas.AccblasError) as info:
        world_of_one.acc_dot(handle, accblas.F16, accblas.F16, accblas.F32, 8, x, 1, x, 1, res)
    assert info.value.status == -1


def test_multi_gpu_driver_on_one_device():
    exe = os.path.join(accblas.PACKAGE_ROOT, "bin", "multi_gpu_benchmark")
    out = subprocess.run([exe, "--gpus=1", "--dot-size=5000001", "--gemv-size=2048",
                          "--runs=3"], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    lines = {l.split(";")[0]: dict(f.split("=") for f in l.strip().split(";")[1:])
             for l in out.splitlines() if l.startswith(("DOT", "GEMV"))}
    dot = lines["DOT Acc<fp32, fp16>"]
    assert dot["all_ranks_identical"] == "yes" and dot["gpus"] == "1"
    # fp32 accumulation in blocks: gate on the natural scale of the sum (the
    # driver's repeated data block makes the sum itself large)
    scale = max(np.sqrt(5000001) / 3, abs(float(dot["expected"])))
    assert float(dot["abs_err"]) <= 64 * 2.0 ** -24 * scale
    gemv = lines["GEMV Acc<fp64, fp32>"]
    assert gemv["all_ranks_identical"] == "yes"
    assert float(gemv["rel_err_sampled_rows"]) <= 1e-7


def test_grouped_dot_casts_after_the_deferred_all_reduce(handle, monkeypatch):
    """Inside accblas_comm_group_begin/_end RCCL only records the all-reduce and
    enqueues it at group end; the cast to a result type other than the
    arithmetic type must still come after it. ACCBLAS_COMM_FORCE_COLLECTIVE=1
    makes the world of one rank run the real ncclAllReduce (partial slot ->
    reduced slot), so a cast enqueued too early would read a stale slot."""
    monkeypatch.setenv("ACCBLAS_COMM_FORCE_COLLECTIVE", "1")
    torch.cuda.set_device(0)
    (c,) = comm.Comm.create_all(1)
    try:
        gen = torch.Generator(device="cpu").manual_seed(21)
        for n, seed_value in [(400_003, 5.0), (1_000_001, -3.0)]:
            x = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
            y = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
            single = torch.zeros(1, dtype=torch.float32, device=DEV)
            accblas.acc_dot(handle, accblas.F32, accblas.F32, accblas.F32, n, x, 1, y, 1, single)
            narrow = torch.full((1,), seed_value, dtype=torch.float32, device=DEV)
            wide = torch.full((1,), seed_value, dtype=torch.float64, device=DEV)
            comm.group_begin()
            c.acc_dot(handle, accblas.F32, accblas.F32, accblas.F64, n, x, 1, y, 1, wide)
            comm.group_end()
            comm.group_begin()
            c.acc_dot(handle, accblas.F32, accblas.F32, accblas.F32, n, x, 1, y, 1, narrow)
            comm.group_end()
            torch.cuda.synchronize()
            assert wide.item() == single.item() and narrow.item() == single.item()
        # ungrouped calls take the same route
        wide = torch.zeros(1, dtype=torch.float64, device=DEV)
        c.acc_dot(handle, accblas.F32, accblas.F32, accblas.F64, n, x, 1, y, 1, wide)
        torch.cuda.synchronize()
        assert wide.item() == single.item()
        # forced all-gather of the one slice
        rows, cols = 128, 512
        A = torch.rand(rows, cols, device=DEV)
        xv = torch.rand(cols, device=DEV)
        local = torch.zeros(rows, device=DEV)
        full = torch.zeros(rows, device=DEV)
        c.acc_gemv(accblas.F64, accblas.F32, rows, cols, 1.0, A, cols, xv, 1, 0.0, local, 1,
                   res_full=full)
        torch.cuda.synchronize()
        assert torch.equal(full, local) and full.abs().sum().item() > 0
    finally:
        c.close()


def test_several_dots_in_one_group_keep_their_own_scalar_slots(monkeypatch):
    """x.y and x.x batched in ONE group on one communicator (what a CG step
    does): inside a group the chunk dots run at once while the all-reduces and
    casts wait for the group end, so every dot needs its own scalar slot (a
    shared slot gave the first result the value of the second). More than 64
    dots in one group are refused."""
    monkeypatch.setenv("ACCBLAS_COMM_FORCE_COLLECTIVE", "1")
    torch.cuda.set_device(0)
    (c,) = comm.Comm.create_all(1)
    h = accblas.Handle()
    try:
        gen = torch.Generator(device="cpu").manual_seed(41)
        n = 700_001
        x = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
        y = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float().to(DEV)
        want = []
        for a, b in [(x, y), (x, x), (y, y)]:
            single = torch.zeros(1, dtype=torch.float32, device=DEV)
            accblas.acc_dot(h, accblas.F32, accblas.F32, accblas.F32, n, a, 1, b, 1, single)
            torch.cuda.synchronize()
            want.append(single.item())
        assert len(set(want)) == 3
        for res_dtype, res_t in [(torch.float64, accblas.F64), (torch.float32, accblas.F32)]:
            results = [torch.full((1,), -1.0, dtype=res_dtype, device=DEV) for _ in range(3)]
            comm.group_begin()
            for (a, b), r in zip([(x, y), (x, x), (y, y)], results):
                c.acc_dot(h, accblas.F32, accblas.F32, res_t, n, a, 1, b, 1, r)
            comm.group_end()
            torch.cuda.synchronize()
            assert [r.item() for r in results] == want
        # the ring has 64 slots: the 65th dot of one group is refused, and the
        # group still closes cleanly with the 64 recorded ones
        results = [torch.zeros(1, dtype=torch.float64, device=DEV) for _ in range(65)]
        comm.group_begin()
        try:
            for r in results[:64]:
                c.acc_dot(h, accblas.F32, accblas.F32, accblas.F64, n, x, 1, y, 1, r)
            with pytest.raises(accblas.AccblasError) as info:
                c.acc_dot(h, accblas.F32, accblas.F32, accblas.F64, n, x, 1, y, 1, results[64])
            assert info.value.status == -2
        finally:
            comm.group_end()
        torch.cuda.synchronize()
        assert all(r.item() == want[0] for r in results[:64])
        # a new group starts with an empty ring again
        comm.group_begin()
        c.acc_dot(h, accblas.F32, accblas.F32, accblas.F64, n, x, 1, x, 1, results[64])
        comm.group_end()
        torch.cuda.synchronize()
        assert results[64].item() == want[1]
    finally:
        h.close()
        c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_two_gpu_grouped_dot_and_gemv(monkeypatch):
    """One process driving two communicators (accblas_comm_create_all): the
    sharded dot (fp32 arithmetic, fp64 result: the cast follows the grouped
    all-reduce) and the row-block GEMV with the all-gather. Runs wherever two
    GPUs are visible (not on the one-GPU test box)."""
    comms = comm.Comm.create_all(2)
    try:
        n = 1_000_003
        gen = torch.Generator(device="cpu").manual_seed(31)
        xs = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float()
        ys = (torch.rand(n, generator=gen, dtype=torch.float64) * 2 - 1).float()
        handles, results, keep = [], [], []
        for c in comms:
            torch.cuda.set_device(c.device)
            handles.append(accblas.Handle())
            results.append(torch.zeros(1, dtype=torch.float64, device=f"cuda:{c.device}"))
        comm.group_begin()
        for c, h, r in zip(comms, handles, results):
            torch.cuda.set_device(c.device)
            lo, hi = c.shard(n, 4)
            x = xs[lo:hi].to(f"cuda:{c.device}")
            y = ys[lo:hi].to(f"cuda:{c.device}")
            keep += [x, y]
            c.acc_dot(h, accblas.F32, accblas.F32, accblas.F64, hi - lo, x, 1, y, 1, r)
        comm.group_end()
        for c in comms:
            torch.cuda.synchronize(c.device)
        want = float((xs.double() * ys.double()).sum())
        scale = float(((xs.double() * ys.double()) ** 2).sum().sqrt())
        assert results[0].item() == results[1].item()
        assert abs(results[0].item() - want) <= 64 * 2.0 ** -24 * scale
        # row-block GEMV (Acc<fp64, fp32>) with the all-gather of the slices:
        # every rank ends with the full result, equal to the single-GPU product
        # bit for bit (row blocks start at multiples of 64 rows)
        rows, cols = 1024, 3000
        A = (torch.rand(rows, cols, generator=gen, dtype=torch.float64) * 2 - 1).float()
        xv = (torch.rand(cols, generator=gen, dtype=torch.float64) * 2 - 1).float()
        r0 = (torch.rand(rows, generator=gen, dtype=torch.float64) * 2 - 1).float()
        torch.cuda.set_device(comms[0].device)
        single = r0.to("cuda:0")
        accblas.acc_gemv(accblas.F64, accblas.F32, rows, cols, 1.0, A.to("cuda:0"), cols,
                         xv.to("cuda:0"), 1, 1.0, single, 1)
        torch.cuda.synchronize(0)
        fulls = []
        comm.group_begin()
        for c in comms:
            torch.cuda.set_device(c.device)
            dev = f"cuda:{c.device}"
            lo, hi = c.shard(rows, 64)
            assert hi - lo == rows // 2
            A_l, x_l, r_l = A[lo:hi].contiguous().to(dev), xv.to(dev), r0[lo:hi].to(dev)
            full = torch.zeros(rows, dtype=torch.float32, device=dev)
            keep += [A_l, x_l, r_l]
            fulls.append(full)
            c.acc_gemv(accblas.F64, accblas.F32, hi - lo, cols, 1.0, A_l, cols, x_l, 1, 1.0,
                       r_l, 1, res_full=full)
        comm.group_end()
        for c in comms:
            torch.cuda.synchronize(c.device)
        for full in fulls:
            assert torch.equal(full.cpu(), single.cpu())
        for h in handles:
            h.close()
    finally:
        for c in comms:
            c.close()
