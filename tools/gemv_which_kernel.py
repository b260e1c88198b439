#!/usr/bin/env python3
"""Developer probe (GPU box, under `rocprofv3 --kernel-trace --stats`): which GEMV
kernel does the library launch for the top-left n x n block of a 24500-wide fp32
matrix, Acc<fp64, fp32>, n = 6000 / 8000 / 9000 / 10000?"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "accessor-blas_amd", "python"))
import accblas  # noqa: E402

accblas.load()
lda, largest = 24500, 10000
A = torch.rand(largest * lda, dtype=torch.float32, device="cuda")
x = torch.rand(largest, dtype=torch.float32, device="cuda")
r = torch.zeros(largest, dtype=torch.float32, device="cuda")
for n in (6000, 8000, 9000, 10000):
    assert n <= largest and n <= lda
    for _ in range(20):
        accblas.acc_gemv(accblas.F64, accblas.F32, n, n, 1.0, A, lda, x, 1, 1.0, r, 1)
    torch.cuda.synchronize()
