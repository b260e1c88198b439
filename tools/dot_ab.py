#!/usr/bin/env python3
"""Developer A/B (GPU box): the dot of the library named by ACCBLAS_LIB_PATH (or
the in-tree build) at a list of sizes, min of 10 event-timed single launches on
the same buffers (the reference's protocol, cuda/utils.cuh:236-262). Run once
per build inside ONE gpurun call, alternating, and compare."""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

if os.environ.get("ACCBLAS_LIB_PATH"):
    accblas.LIB_PATH = os.environ["ACCBLAS_LIB_PATH"]
F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64
h = accblas.Handle()


def t(n, ar, st):
    x = torch.rand(n, dtype=torch.float32, device="cuda").to(accblas.torch_dtype(st))
    y = torch.rand(n, dtype=torch.float32, device="cuda").to(accblas.torch_dtype(st))
    res = torch.zeros(1, dtype=accblas.torch_dtype(ar), device="cuda")
    fn = lambda: accblas.acc_dot(h, ar, st, ar, n, x, 1, y, 1, res)
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(10):
        a, b = accblas.Event(), accblas.Event()
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b))
    print(f"{os.environ.get('ACCBLAS_LIB_PATH', 'in-tree'):>44s} n={n:>10d} "
          f"Acc<{accblas.DTYPE_NAMES[ar]},{accblas.DTYPE_NAMES[st]}>: {best * 1e3:8.2f} us "
          f"{2 * accblas.DTYPE_BYTES[st] * n / best / 1e6:8.1f} GB/s", flush=True)


for n in (1000, 100_000, 1_000_000, 3_000_000, 10_000_000):
    t(n, F64, F64)
for n in (1_000_000, 1 << 24, 1 << 25, 1 << 26):
    t(n, F32, F16)
for n in (1 << 26, 1 << 28):
    t(n, F64, F32)
t(1 << 28, F32, F16)
