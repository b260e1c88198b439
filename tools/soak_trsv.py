#!/usr/bin/env python3
"""Developer soak (GPU box): thousands of back-to-back solves of random sizes,
uplo, diag and storage types on two streams, checking the helper word and a
residual every so often. Exercises the self re-arming workspace protocol."""
import os
import sys

import numpy as np
import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64


def main(iterations=4000, seed=7, max_n=14000):
    # (14000: 110 row blocks of 128, the split blocks can be exercised; from
    # ~17000 on the 64-row kernel has more row blocks than CUs and its workgroups
    # draw tickets until none is left: second argument)
    rng = np.random.default_rng(seed)
    scale = 1.0 / np.sqrt(max_n)
    base = (torch.rand(max_n * max_n, dtype=torch.float64, device="cuda") * 2 - 1) * scale
    mats = {F64: base, F32: base.float(), F16: base.half()}
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    helpers = [torch.zeros(2, dtype=torch.int32, device="cuda") for _ in streams]
    worst = 0.0
    for it in range(iterations):
        n = int(rng.integers(1, max_n + 1))
        st = (F64, F32, F16)[int(rng.integers(0, 3))]
        ar = F64 if st == F64 or rng.integers(0, 2) else F32
        upper = bool(rng.integers(0, 2))
        which = it % 2
        # kernel family: automatic / 64-row only / 128-row wherever possible,
        # the latter now and then with split row blocks (helpers from block 98)
        family = int(rng.integers(0, 4))
        accblas.set_option("trsv_wide", (1, 0, 2, 2)[family])
        accblas.set_option("trsv_wide_split_from", 98 if family == 3 else 0)
        # (solves with more tickets than CUs: a workgroup per ticket, or one per
        # CU drawing tickets - the two forms share a workspace and an epoch)
        accblas.set_option("trsv_wide_persistent", int(rng.integers(0, 2)))
        # every third solve on a view that is not 16-byte aligned: a base a few
        # elements into the allocation and an odd row stride
        off, lda = (int(rng.integers(1, 8)), max_n - 1 - 2 * int(rng.integers(0, 3))) if it % 3 == 2 else (0, max_n)
        n = min(n, lda - 8)
        A = mats[st][off:]
        dtype = accblas.torch_dtype(st)
        with torch.cuda.stream(streams[which]):
            # operands are produced on the stream that consumes them
            b = (torch.rand(n, dtype=torch.float64, device="cuda") * 2 - 1).to(dtype)
            x = b.clone()
            accblas.acc_trsv(ar, st, accblas.UPPER if upper else accblas.LOWER,
                             accblas.UNIT, n, A, lda, x, 1, helpers[which], streams[which])
        if it % 200 == 199:
            torch.cuda.synchronize()
            for h in helpers:
                assert int(h[0].item()) & 0xffffffff == 0xffffffff, f"spin expired at {it}"
            # residual of the last solve: T x = b with T = unit triangle of A[:n, :n]
            T = torch.as_strided(A, (n, n), (lda, 1)).double()
            T = torch.triu(T, 1) if upper else torch.tril(T, -1)
            T = T + torch.eye(n, dtype=torch.float64, device="cuda")
            res = (T @ x.double() - b.double()).abs().sum() / b.double().abs().sum()
            worst = max(worst, float(res))
            tol = {F64: 1e-12, F32: 1e-5, F16: 2e-2}[st]
            assert float(res) <= tol, (it, n, st, ar, upper, float(res))
            print(f"iteration {it + 1}: ok (n={n}, lda={lda}, base+{off}, st={accblas.DTYPE_NAMES[st]}, "
                  f"residual {float(res):.2e})", flush=True)
    torch.cuda.synchronize()
    accblas.set_option("trsv_wide", 1)
    accblas.set_option("trsv_wide_split_from", 0)
    accblas.set_option("trsv_wide_persistent", 0)
    print(f"soak passed: {iterations} solves, worst checked residual {worst:.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4000,
         max_n=int(sys.argv[2]) if len(sys.argv) > 2 else 14000)
