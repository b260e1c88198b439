#!/usr/bin/env python3
"""Developer probe (GPU box): 64-row kernel vs 128-row kernel (option
"trsv_wide" = 0 / 2) over sizes and type pairs, alternated in ONE process so
that both see the same box. usage: trsv_ab.py [sizes...]"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

if os.environ.get("ACCBLAS_LIB_PATH"):  # A/B runs against another build
    accblas.LIB_PATH = os.environ["ACCBLAS_LIB_PATH"]

F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64


def time_us(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = accblas.Event(), accblas.Event()
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_ms(b) * 1e3)
    return best


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 24000, 32768]
    print(accblas.device_info(0), flush=True)
    for n in sizes:
        scale = 1.0 / n ** 0.5
        base = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale)
        xb = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1)
        helper = torch.zeros(2, dtype=torch.int32, device="cuda")
        for ar, st in [(F64, F32), (F64, F64), (F32, F32), (F64, F16), (F32, F16)]:
            if n > 40000 and st == F64:
                continue
            A = base.to(accblas.torch_dtype(st))
            x0 = xb.to(accblas.torch_dtype(st))
            x = x0.clone()
            for upper in (False, True):
                row = []
                for wide in (0, 2, 0, 2):
                    accblas.set_option("trsv_wide", wide)

                    def solve():
                        x.copy_(x0)
                        accblas.acc_trsv(ar, st, accblas.UPPER if upper else accblas.LOWER,
                                         accblas.UNIT, n, A, n, x, 1, helper)
                    row.append(time_us(solve, 5 if n > 30000 else 8))
                t64, t128 = min(row[0], row[2]), min(row[1], row[3])
                byts = accblas.DTYPE_BYTES[st] * (n * (n - 1) / 2 + 2 * n)
                print(f"n={n:6d} Acc<{accblas.DTYPE_NAMES[ar]},{accblas.DTYPE_NAMES[st]}> "
                      f"{'upper' if upper else 'lower'}: rows64 {t64:9.1f} us  rows128 {t128:9.1f} us  "
                      f"ratio {t128 / t64:5.2f}  rows128 {byts / t128 / 1e3:7.1f} GB/s "
                      f"helper={helper[0].item() & 0xffffffff:08x}", flush=True)
            del A
    accblas.set_option("trsv_wide", 1)


if __name__ == "__main__":
    main()
