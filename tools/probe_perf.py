#!/usr/bin/env python3
"""Developer probe (GPU box): min-of-10 event timings of the three operations at
the BASELINE sizes, printed as GB/s against the algorithmic byte counts of
SURVEY.md §8d D3. Not the contract benchmark (that is bench.py)."""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

if os.environ.get("ACCBLAS_LIB_PATH"):  # A/B runs against another build
    accblas.LIB_PATH = os.environ["ACCBLAS_LIB_PATH"]

F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64
NAME = accblas.DTYPE_NAMES
SIZE = accblas.DTYPE_BYTES


def time_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = accblas.Event(), accblas.Event()
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b))
    return best


def rand(n, dt, scale=1.0):
    t = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale)
    return t.to(accblas.torch_dtype(dt))


def gemv(n, ar, st, plain=False, incx=1):
    A, x, r = rand(n * n, st), rand(n * incx, st), rand(n, st)
    if plain:
        fn = lambda: accblas.gemv(st, n, n, 1.0, A, n, x, incx, 1.0, r, 1)
    else:
        fn = lambda: accblas.acc_gemv(ar, st, n, n, 1.0, A, n, x, incx, 1.0, r, 1)
    ms = time_ms(fn)
    byts = SIZE[st] * (n * n + 3 * n)
    print(f"GEMV n={n} incx={incx} {'plain ' + NAME[st] if plain else 'Acc<' + NAME[ar] + ',' + NAME[st] + '>'}"
          f": {ms * 1e3:9.1f} us  {byts / ms / 1e6:8.1f} GB/s  {2 * n * n / ms / 1e6:8.1f} GFLOP/s",
          flush=True)


def dot(n, ar, st, mode=0):
    h = accblas.Handle()
    h.set_dot_mode(mode)
    x, y = rand(n, st), rand(n, st)
    res = torch.zeros(1, dtype=accblas.torch_dtype(ar), device="cuda")
    ms = time_ms(lambda: accblas.acc_dot(h, ar, st, ar, n, x, 1, y, 1, res))
    byts = 2 * SIZE[st] * n
    print(f"DOT  n={n} Acc<{NAME[ar]},{NAME[st]}> mode={mode}: {ms * 1e3:9.1f} us  "
          f"{byts / ms / 1e6:8.1f} GB/s", flush=True)
    h.close()


def trsv(n, ar, st, upper, unit=True):
    scale = 1.0 / (n ** 0.5)
    A = rand(n * n, st, scale)
    if not unit:
        A.view(n, n).diagonal().fill_(1.5)
    x0 = rand(n, st)
    x = x0.clone()
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    uplo = accblas.UPPER if upper else accblas.LOWER

    def fn():
        x.copy_(x0)
        accblas.acc_trsv(ar, st, uplo, accblas.UNIT if unit else accblas.NON_UNIT, n, A, n,
                         x, 1, helper)
    ms = time_ms(fn, reps=5)
    byts = SIZE[st] * (n * (n - 1) / 2 + 2 * n)
    print(f"TRSV n={n} {'upper' if upper else 'lower'} Acc<{NAME[ar]},{NAME[st]}>: "
          f"{ms * 1e3:9.1f} us  {byts / ms / 1e6:8.1f} GB/s  finite={bool(torch.isfinite(x.float()).all())}",
          flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    print(accblas.device_info(0), flush=True)
    if os.environ.get("ACCBLAS_TRSV_WIDE"):
        accblas.set_option("trsv_wide", int(os.environ["ACCBLAS_TRSV_WIDE"]))
        print("trsv_wide =", os.environ["ACCBLAS_TRSV_WIDE"], flush=True)
    if what in ("all", "gemv"):
        gemv(16384, F64, F64, plain=True)
        gemv(16384, F64, F64)
        gemv(32768, F64, F32)
        gemv(32768, F32, F32)
        gemv(32768, F64, F16)
        gemv(32768, F32, F16)
        gemv(24500, F64, F32)
        gemv(32767, F64, F32)  # odd row stride: rows are only 4-byte aligned
        gemv(32766, F64, F16)  # fp16 rows 4-byte aligned
        gemv(32768, F64, F32, incx=2)
        gemv(32768, F64, F16, incx=3)
        gemv(4096, F64, F32)
        gemv(1024, F64, F32)
    if what in ("all", "dot"):
        for mode in (0, 1):
            dot(1 << 28, F64, F32, mode)
            dot(1 << 28, F32, F16, mode)
            dot(1 << 25, F32, F16, mode)
            dot(1 << 28, F64, F64, mode)
            dot(1_000_000, F64, F64, mode)
    if what in ("all", "trsv"):
        for n in (1024, 8192, 24000, 65536):
            trsv(n, F64, F32, upper=False)
        trsv(24000, F64, F32, upper=True)
        trsv(24000, F64, F64, upper=True)
        trsv(24000, F32, F32, upper=True)
