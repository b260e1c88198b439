#!/usr/bin/env python3
"""Developer probe (GPU box): the row blocks of the strong-scaling GEMV
(32768 columns, rows = 32768 / N for N = 1, 2, 4, 8) on one GPU: event-timed
min of 10 single launches and the per-launch average of 20 back-to-back ones."""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

F32, F64 = accblas.F32, accblas.F64


def main():
    cols = lda = 32768
    A = torch.rand(32768 * lda, dtype=torch.float32, device="cuda").mul_(2).sub_(1)
    x = torch.rand(cols, dtype=torch.float32, device="cuda")
    r = torch.zeros(32768, dtype=torch.float32, device="cuda")
    for skew in (1, 0):
        accblas.set_option("gemv_skew", skew)
        for rows in (32768, 16384, 8192, 4096, 2048):
            fn = lambda: accblas.acc_gemv(F64, F32, rows, cols, 1.0, A, lda, x, 1, 1.0, r, 1)
            fn()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(10):
                a, b = accblas.Event(), accblas.Event()
                a.record(); fn(); b.record(); b.synchronize()
                best = min(best, a.elapsed_ms(b) * 1e3)
            batch = 1e30
            for _ in range(5):
                a, b = accblas.Event(), accblas.Event()
                a.record()
                for _ in range(20):
                    fn()
                b.record(); b.synchronize()
                batch = min(batch, a.elapsed_ms(b) * 1e3 / 20)
            byts = 4.0 * (rows * cols + 2 * rows + cols)
            print(f"gemv_skew={skew} rows={rows:6d}: single {best:7.1f} us {byts / best / 1e3:7.1f} GB/s | "
                  f"back-to-back {batch:7.1f} us {byts / batch / 1e3:7.1f} GB/s", flush=True)
    accblas.set_option("gemv_skew", 1)


if __name__ == "__main__":
    main()
