#!/usr/bin/env python3
"""Developer probe (GPU box): is `trsv<T>` faster than the bit-identical
`acc_trsv<T, T>`, or does the benchmark's ORDER make it look so?

The two kernels have the same instruction stream (ISA diff: only the kernel
argument loads differ), so this script separates kernel from context. Upper /
unit (the CLI's default), top-left n x n block of an lda-wide matrix.
  A  alternate plain / accessor solve by solve, x restored before each
     (outside the timed region)
  B  the CLI's protocol: a column = warm-up + minimum of 10 back-to-back solves
     WITHOUT restoring x (`benchmark_function`), columns in the CLI's order
     plain T, [other column], Acc<T,T>; then in the opposite order
  C  protocol B with x restored before every solve
  D  one column on an x that holds NaN / Inf / denormals / zeros
usage: trsv_plain_vs_acc.py [f64|f32] [sizes...]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "accessor-blas_amd", "python"))
import accblas  # noqa: E402

accblas.load()
T = {"f64": (accblas.F64, torch.float64), "f32": (accblas.F32, torch.float32)}
name = sys.argv[1] if len(sys.argv) > 1 else "f64"
dt, tdt = T[name]
sizes = [int(a) for a in sys.argv[2:]] or [22100, 23200, 24000]
LDA = int(os.environ.get("LDA", "24000"))
helper = torch.zeros(2, dtype=torch.int32, device="cuda")


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3


def fmt(ts):
    return f"min {min(ts):7.1f} med {statistics.median(ts):7.1f} max {max(ts):7.1f}"


print(f"T={name} upper/unit lda={LDA}", flush=True)
big = torch.rand(24000 * LDA, dtype=torch.float32, device="cuda")
for n in sizes:
    scale = 1.0 / n ** 0.5
    A = big.mul(2 * scale).sub_(scale).to(tdt)
    A32 = A.to(torch.float32) if tdt == torch.float64 else None
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(tdt)
    x = x0.clone()
    x32 = x0.to(torch.float32)

    def plain():
        accblas.trsv(dt, accblas.UPPER, accblas.UNIT, n, A, LDA, x, 1, helper)

    def acc():
        accblas.acc_trsv(dt, dt, accblas.UPPER, accblas.UNIT, n, A, LDA, x, 1, helper)

    def other():  # the column between the two in the CLI (fp32 storage, 128-row kernel)
        if A32 is not None:
            accblas.trsv(accblas.F32, accblas.UPPER, accblas.UNIT, n, A32, LDA, x32, 1, helper)

    # A
    tp, ta = [], []
    for rep in range(13):
        for which, fn, out in ((0, plain, tp), (1, acc, ta)):
            x.copy_(x0)
            torch.cuda.synchronize()
            t = timed(fn)
            if rep:
                out.append(t)
    print(f"n={n} A alternating, x restored: plain {fmt(tp)} | acc {fmt(ta)} | "
          f"ratio(min) {min(ta) / min(tp):.3f} ratio(med) "
          f"{statistics.median(ta) / statistics.median(tp):.3f}", flush=True)

    def column(fn, restore):
        if restore:
            x.copy_(x0)
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            if restore:
                x.copy_(x0)
                torch.cuda.synchronize()
            ts.append(timed(fn))
        return ts

    for restore, label in ((False, "B x not restored"), (True, "C x restored")):
        for order in ("plain,other,acc", "acc,other,plain"):
            x.copy_(x0)
            got = {}
            for _round in range(3):
                for col in order.split(","):
                    ts = column({"plain": plain, "acc": acc, "other": other}[col], restore)
                    got.setdefault(col, []).append(ts)
            mins = {c: [min(t) for t in v] for c, v in got.items()}
            print(f"n={n} {label}, order {order}: plain mins "
                  f"{' '.join(f'{m:.1f}' for m in mins['plain'])} | acc mins "
                  f"{' '.join(f'{m:.1f}' for m in mins['acc'])} | all plain "
                  f"{' '.join(f'{t:.0f}' for t in got['plain'][0])} | all acc "
                  f"{' '.join(f'{t:.0f}' for t in got['acc'][0])} | x finite: "
                  f"{bool(torch.isfinite(x).all().item())}", flush=True)

    # D
    for label, value in (("nan", float("nan")), ("inf", float("inf")), ("denormal", 1e-310),
                         ("zero", 0.0), ("one", 1.0)):
        ts = []
        for _ in range(8):
            x.fill_(value)
            torch.cuda.synchronize()
            ts.append(timed(plain))
        print(f"n={n} D x = {label}: plain {fmt(ts)}", flush=True)
    del A, A32
