#!/usr/bin/env python3
"""Developer probe (GPU box): runs ONE TRSV configuration a few times so that a
profiler wrapped around it sees only that kernel.
usage: trsv_case.py N [ar st uplo reps relay wide]   e.g. trsv_case.py 65536 f64 f32 lower 5 0 0"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

if os.environ.get("ACCBLAS_LIB_PATH"):
    accblas.LIB_PATH = os.environ["ACCBLAS_LIB_PATH"]
T = {"f16": accblas.F16, "f32": accblas.F32, "f64": accblas.F64}


def main():
    n = int(sys.argv[1])
    ar = T[sys.argv[2]] if len(sys.argv) > 2 else accblas.F64
    st = T[sys.argv[3]] if len(sys.argv) > 3 else accblas.F32
    upper = len(sys.argv) > 4 and sys.argv[4] == "upper"
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    relay = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    wide = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    accblas.set_option("trsv_relay", relay)
    if wide >= 0:
        accblas.set_option("trsv_wide", wide)
    scale = 1.0 / n ** 0.5
    A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale)
    A = A.to(accblas.torch_dtype(st))
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(accblas.torch_dtype(st))
    x = x0.clone()
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    best = 1e30
    for _ in range(reps + 1):
        x.copy_(x0)
        a, b = accblas.Event(), accblas.Event()
        a.record()
        accblas.acc_trsv(ar, st, accblas.UPPER if upper else accblas.LOWER, accblas.UNIT, n, A, n,
                         x, 1, helper)
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b))
    byts = accblas.DTYPE_BYTES[st] * (n * (n - 1) / 2 + 2 * n)
    print(f"TRSV n={n} Acc<{accblas.DTYPE_NAMES[ar]},{accblas.DTYPE_NAMES[st]}> "
          f"{'upper' if upper else 'lower'} relay={relay} wide={wide}: {best * 1e3:9.1f} us "
          f"{byts / best / 1e6:8.1f} GB/s algorithmic_bytes={byts:.0f} "
          f"helper0={helper[0].item() & 0xffffffff:08x}", flush=True)


if __name__ == "__main__":
    main()
