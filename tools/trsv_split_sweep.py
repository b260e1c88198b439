#!/usr/bin/env python3
"""Developer probe (GPU box): the 128-row TRSV kernel with row blocks split
into parts from different first blocks (option "trsv_wide_split_from"),
alternated in ONE process. usage: trsv_split_sweep.py [sizes...]"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

from trsv_ab import time_us  # noqa: E402

FIRSTS = [0, 288, 320, 352, 384, 448]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16384, 24000, 32768, 40000, 65536]
    print(accblas.device_info(0), flush=True)
    accblas.set_option("trsv_wide", 2)
    for n in sizes:
        scale = 1.0 / n ** 0.5
        A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale)
        x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1)
        x = x0.clone()
        helper = torch.zeros(2, dtype=torch.int32, device="cuda")
        best = {f: 1e30 for f in FIRSTS}
        bits = {}
        for _ in range(2):
            for first in FIRSTS:
                accblas.set_option("trsv_wide_split_from", first)

                def solve():
                    x.copy_(x0)
                    accblas.acc_trsv(accblas.F64, accblas.F32, accblas.LOWER, accblas.UNIT,
                                     n, A, n, x, 1, helper)
                best[first] = min(best[first], time_us(solve, 4))
                bits[first] = x.clone()
        ref = bits[0].double()
        line = "  ".join(
            f"{f or 'cus'}: {best[f]:7.1f} ({(bits[f].double() - ref).abs().sum().item() / ref.abs().sum().item():.1e})"
            for f in FIRSTS)
        print(f"n={n:6d} Acc<f64,f32> lower/unit  first split block -> us (rel. diff to unsplit-first result)\n   {line}"
              f"  helper={helper[0].item() & 0xffffffff:08x}", flush=True)
        del A
    accblas.set_option("trsv_wide_split_from", 0)
    accblas.set_option("trsv_wide", 1)


if __name__ == "__main__":
    main()
