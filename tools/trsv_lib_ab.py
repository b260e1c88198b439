#!/usr/bin/env python3
"""Developer A/B (GPU box): acc_trsv of the library named by ACCBLAS_LIB_PATH
(or the in-tree build), lower / unit, min of 8 event-timed solves per size
(x restored by a device copy inside the timed region, as bench.py does; the
copy's own time is printed once). Run once per build inside ONE gpurun call,
alternating, and compare. usage: trsv_lib_ab.py [st=f32] [sizes...]"""
import os
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

if os.environ.get("ACCBLAS_LIB_PATH"):
    accblas.LIB_PATH = os.environ["ACCBLAS_LIB_PATH"]
F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64
label = os.path.basename(os.path.dirname(os.environ.get("ACCBLAS_LIB_PATH", "/lib/x")))


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
ar = F64
sizes = [int(a) for a in args] or [4096, 8192, 16384, 24000, 32768, 40000, 49152, 65536]
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
out = []
for n in sizes:
    scale = 1.0 / n ** 0.5
    A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale) \
        .to(accblas.torch_dtype(st))
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(accblas.torch_dtype(st))
    x = x0.clone()

    def solve():
        x.copy_(x0)
        accblas.acc_trsv(ar, st, accblas.LOWER, accblas.UNIT, n, A, n, x, 1, helper)
    t_copy = time_us(lambda: x.copy_(x0), 8)
    t = time_us(solve, 8) - t_copy
    ok = helper[0].item() & 0xffffffff
    out.append(f"{n}:{t:8.1f}")
    del A
print(f"{label:>12s} st={accblas.DTYPE_NAMES[st]} " + "  ".join(out) + f"  helper={ok:08x}", flush=True)
