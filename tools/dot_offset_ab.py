#!/usr/bin/env python3
"""Developer A/B (GPU box): acc_dot of subvectors that start OFFSET elements
into their allocations, several builds of libaccblas_hip.so alternated in ONE
process (min / median of `reps` event-timed launches). Acc<fp64, ST>.
usage: dot_offset_ab.py reps lib_dir[,lib_dir...] x_offset y_offset [n ...]   (ST=f16|f32|f64 in the environment)"""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ST, ST_TORCH = {"f16": (0, torch.float16), "f32": (1, torch.float32), "f64": (2, torch.float64)}[
    os.environ.get("ST", "f32")]
F64 = 2
reps, names = int(sys.argv[1]), sys.argv[2].split(",")
x_off, y_off = int(sys.argv[3]), int(sys.argv[4])
sizes = [int(a) for a in sys.argv[5:]] or [1_000_000, 1 << 24, 1 << 26]
i64, vp = ctypes.c_int64, ctypes.c_void_p
libs, handles = [], []
for name in names:
    lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", name, "libaccblas_hip.so"))
    lib.accblas_handle_create.restype = ctypes.c_int
    lib.accblas_handle_create.argtypes = [ctypes.POINTER(vp)]
    lib.accblas_acc_dot.restype = ctypes.c_int
    lib.accblas_acc_dot.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, vp, i64, vp, i64, vp, vp]
    handle = vp()
    assert lib.accblas_handle_create(ctypes.byref(handle)) == 0
    libs.append(lib)
    handles.append(handle)
for n in sizes:
    x = torch.rand(n + x_off, dtype=torch.float32, device="cuda").to(ST_TORCH)[x_off:]
    y = torch.rand(n + y_off, dtype=torch.float32, device="cuda").to(ST_TORCH)[y_off:]
    res = torch.zeros(1, dtype=torch.float64, device="cuda")

    def timed(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = libs[k].accblas_acc_dot(handles[k], F64, ST, F64, n, x.data_ptr(), 1, y.data_ptr(), 1,
                                     res.data_ptr(), None)
        b.record()
        b.synchronize()
        assert rc == 0, rc
        return a.elapsed_time(b) * 1e3
    times = [[] for _ in libs]
    for rep in range(-2, reps):
        for k in range(len(libs)):
            i = (k + rep) % len(libs)
            t = timed(i)
            if rep >= 0:
                times[i].append(t)
    print(f"n={n:>10d} offsets {x_off},{y_off}: " + "  ".join(
        f"{names[i]}: {min(t):7.1f} / {statistics.median(t):7.1f}" for i, t in enumerate(times)), flush=True)
