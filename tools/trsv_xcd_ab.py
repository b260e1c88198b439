#!/usr/bin/env python3
"""Developer A/B (GPU box): acc_trsv Acc<fp64, st>, lower / unit, with option
"trsv_wide_xcd_run" = 0 (row blocks in ticket order) against runs of k row
blocks per XCD, alternated inside one process, min of 8 event-timed solves per
setting and size (x restored by a device copy inside the timed region, the
copy's own time subtracted). usage: trsv_xcd_ab.py [st=f32] [sizes...]"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

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


args = sys.argv[1:]
st = F32
if args and args[0] in ("f16", "f32", "f64"):
    st = {"f16": F16, "f32": F32, "f64": F64}[args.pop(0)]
sizes = [int(a) for a in args] or [4096, 8192, 16384, 24000, 32768, 49152, 65536]
runs = [0, 1, 2, 4, 8, 16, 0]
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
print("xcd_run " + " ".join(f"{n:>8d}" for n in sizes), flush=True)
table = {r: [] for r in range(len(runs))}
for n in sizes:
    scale = 1.0 / n ** 0.5
    A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale) \
        .to(accblas.torch_dtype(st))
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(accblas.torch_dtype(st))
    x = x0.clone()

    def solve():
        x.copy_(x0)
        accblas.acc_trsv(F64, st, accblas.LOWER, accblas.UNIT, n, A, n, x, 1, helper)
    t_copy = time_us(lambda: x.copy_(x0), 8)
    first = None
    for i, run in enumerate(runs):
        accblas.set_option("trsv_wide_xcd_run", run)
        t = time_us(solve, 8) - t_copy
        assert helper[0].item() & 0xffffffff == 0xffffffff
        if first is None:
            first = x.clone()
        assert torch.equal(first, x), "placement changed the bits"
        table[i].append(t)
    accblas.set_option("trsv_wide_xcd_run", 0)
    del A
for i, run in enumerate(runs):
    print(f"{run:7d} " + " ".join(f"{t:8.1f}" for t in table[i]), flush=True)
