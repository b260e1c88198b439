#!/usr/bin/env python3
"""Developer A/B (GPU box): accblas_acc_gemv of SEVERAL builds of
libaccblas_hip.so in ONE process on the same buffers, launch by launch in
rotating order. Acc<fp64, fp32> by default; min and median of `reps`
event-timed launches per build and shape.
usage: gemv_lib_ab2.py reps lib_dir[,lib_dir...] [rows:cols:lda ...]   (lib_dir under accessor-blas_amd/;
lib_dir:option=value sets an accblas option in that build - two settings of one build need two copies of its
directory; ST=f16|f32|f64 in the environment: storage type, Acc<fp64, ST>; OFFSET=k: A and x start k elements
into their allocations)"""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
F32, F64 = 1, 2
ST, ST_TORCH = {"f16": (0, torch.float16), "f32": (1, torch.float32), "f64": (2, torch.float64)}[
    os.environ.get("ST", "f32")]
OFFSET = int(os.environ.get("OFFSET", "0"))
reps = int(sys.argv[1])
names = sys.argv[2].split(",")
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[3:]] or [(32768, 32768, 32768)]
i64, vp, dbl = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
libs = []
for name in names:
    directory, _, setting = name.partition(":")
    lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", directory, "libaccblas_hip.so"))
    for one in filter(None, setting.split("+")):
        option, _, value = one.partition("=")
        lib.accblas_set_option.restype = ctypes.c_int
        lib.accblas_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        assert lib.accblas_set_option(option.encode(), int(value)) == 0, one
    lib.accblas_acc_gemv.restype = ctypes.c_int
    lib.accblas_acc_gemv.argtypes = [ctypes.c_int, ctypes.c_int, i64, i64, dbl, vp, i64, vp, i64, dbl, vp,
                                     i64, vp]
    libs.append(lib)
for rows, cols, lda in shapes:
    A = torch.rand(rows * lda + OFFSET, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(ST_TORCH)[OFFSET:]
    x = torch.rand(cols + OFFSET, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(ST_TORCH)[OFFSET:]
    y = torch.zeros(rows, dtype=ST_TORCH, device="cuda")

    def timed(lib):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.accblas_acc_gemv(F64, ST, rows, cols, 1.0, A.data_ptr(), lda, x.data_ptr(), 1, 0.0,
                                  y.data_ptr(), 1, None)
        b.record()
        b.synchronize()
        assert rc == 0, rc
        return a.elapsed_time(b) * 1e3
    times = [[] for _ in libs]
    for rep in range(-2, reps):
        for k in range(len(libs)):
            i = (k + rep) % len(libs)
            t = timed(libs[i])
            if rep >= 0:
                times[i].append(t)
    # the same launches back to back (what bench.py times): mean of a batch
    batch = []
    for lib in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            lib.accblas_acc_gemv(F64, ST, rows, cols, 1.0, A.data_ptr(), lda, x.data_ptr(), 1, 0.0,
                                 y.data_ptr(), 1, None)
        b.record()
        b.synchronize()
        batch.append(a.elapsed_time(b) * 1e3 / 50)
    print("   50 launches back to back, us per launch: " + "  ".join(
        f"{names[i]}: {t:7.1f}" for i, t in enumerate(batch)), flush=True)
    print(f"{rows}x{cols} lda {lda}: " + "  ".join(
        f"{names[i]}: {min(t):7.1f} / {statistics.median(t):7.1f}" for i, t in enumerate(times)), flush=True)
    del A
