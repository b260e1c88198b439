#!/usr/bin/env python3
"""Developer probe (GPU box): runs ONE dot configuration a few times so that a
profiler wrapped around it sees only that kernel.
usage: dot_case.py N [ar st reps]   e.g. dot_case.py 268435456 f32 f16 5"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

T = {"f16": accblas.F16, "f32": accblas.F32, "f64": accblas.F64}


def main():
    n = int(sys.argv[1])
    ar = T[sys.argv[2]] if len(sys.argv) > 2 else accblas.F32
    st = T[sys.argv[3]] if len(sys.argv) > 3 else accblas.F16
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    x = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(accblas.torch_dtype(st))
    y = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(accblas.torch_dtype(st))
    res = torch.zeros(1, dtype=accblas.torch_dtype(ar), device="cuda")
    handle = accblas.Handle()
    best = 1e30
    for _ in range(reps + 1):
        a, b = accblas.Event(), accblas.Event()
        a.record()
        accblas.acc_dot(handle, ar, st, ar, n, x, 1, y, 1, res)
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b))
    byts = 2 * accblas.DTYPE_BYTES[st] * n
    print(f"DOT n={n} Acc<{accblas.DTYPE_NAMES[ar]},{accblas.DTYPE_NAMES[st]}>: {best * 1e3:9.1f} us "
          f"{byts / best / 1e6:8.1f} GB/s algorithmic_bytes={byts:.0f}", flush=True)
    handle.close()


if __name__ == "__main__":
    main()
