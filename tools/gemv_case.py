#!/usr/bin/env python3
"""Developer probe (GPU box): runs ONE GEMV configuration a few times so that a
profiler wrapped around it sees only that kernel.
usage: gemv_case.py ROWS COLS [ar st reps lda]   e.g. gemv_case.py 16384 16384 f64 f64 5
       (lda defaults to COLS; a row block of a wider matrix: gemv_case.py 4096 32768 f64 f32 5)"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

T = {"f16": accblas.F16, "f32": accblas.F32, "f64": accblas.F64}


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    ar = T[sys.argv[3]] if len(sys.argv) > 3 else accblas.F64
    st = T[sys.argv[4]] if len(sys.argv) > 4 else accblas.F32
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    lda = int(sys.argv[6]) if len(sys.argv) > 6 else cols
    dt = accblas.torch_dtype(st)
    A = torch.rand(rows * lda, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(dt)
    x = torch.rand(cols, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(dt)
    res = torch.zeros(rows, dtype=dt, device="cuda")
    best = 1e30
    for _ in range(reps + 1):
        a, b = accblas.Event(), accblas.Event()
        a.record()
        accblas.acc_gemv(ar, st, rows, cols, 1.0, A, lda, x, 1, 1.0, res, 1)
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b))
    byts = accblas.DTYPE_BYTES[st] * (rows * cols + 2.0 * rows + cols)
    print(f"GEMV {rows}x{cols} Acc<{accblas.DTYPE_NAMES[ar]},{accblas.DTYPE_NAMES[st]}>: "
          f"{best * 1e3:9.1f} us {byts / best / 1e6:8.1f} GB/s algorithmic_bytes={byts:.0f}", flush=True)


if __name__ == "__main__":
    main()
