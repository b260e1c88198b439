#!/usr/bin/env python3
"""Developer probe (GPU box): 64-row vs 128-row TRSV kernel for fp64 storage
(option "trsv_wide" = 0 / 2), alternated in one process.
usage: trsv_f64_ab.py [lda] [sizes...]   (lda = 0: row stride = n)"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

from trsv_ab import time_us  # noqa: E402

F64 = accblas.F64


def main():
    lda_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    sizes = [int(a) for a in sys.argv[2:]] or [16384, 24000, 32768, 49152, 65536]
    for n in sizes:
        lda = lda_arg if lda_arg >= n else n
        s = 1 / n ** 0.5
        A = torch.rand(n * lda, dtype=torch.float32, device="cuda").mul_(2 * s).sub_(s).double()
        x0 = torch.rand(n, dtype=torch.float64, device="cuda") * 2 - 1
        x = x0.clone()
        h = torch.zeros(2, dtype=torch.int32, device="cuda")
        r = []
        for w in (0, 2, 0, 2):
            accblas.set_option("trsv_wide", w)

            def solve():
                x.copy_(x0)
                accblas.acc_trsv(F64, F64, accblas.LOWER, accblas.UNIT, n, A, lda, x, 1, h)
            r.append(time_us(solve, 5))
        t64, t128 = min(r[0], r[2]), min(r[1], r[3])
        print(f"n={n} lda={lda} fp64/fp64 rows64 {t64:.1f} rows128 {t128:.1f} ratio {t128 / t64:.3f} "
              f"helper={h[0].item() & 0xffffffff:08x}", flush=True)
        del A
    accblas.set_option("trsv_wide", 1)


if __name__ == "__main__":
    main()
