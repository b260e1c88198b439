#!/usr/bin/env python3
"""Developer probe (GPU box): host time per accblas_acc_trsv call (1000 calls
enqueued without synchronising) and event time per solve for small systems,
for several builds in one process. usage: trsv_small_overheads.py lib_dir[,lib_dir...] [sizes...]"""
import ctypes
import os
import sys
import time

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
names = sys.argv[1].split(",")
sizes = [int(a) for a in sys.argv[2:]] or [100, 200, 500, 1000, 2000]
i64, vp = ctypes.c_int64, ctypes.c_void_p
libs = []
for name in names:
    lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", name, "libaccblas_hip.so"))
    lib.accblas_acc_trsv.restype = ctypes.c_int
    lib.accblas_acc_trsv.argtypes = [ctypes.c_int] * 4 + [i64, vp, i64, vp, i64, vp, vp]
    libs.append(lib)
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
# the floor under every number below: the same three measurements of a kernel
# that does nothing but one 4-byte store
word = torch.zeros(4, dtype=torch.int32, device="cuda")
for _ in range(20):
    word[:1].zero_()
torch.cuda.synchronize()
best = 1e9
for _ in range(50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    word[:1].zero_()
    b.record()
    b.synchronize()
    best = min(best, a.elapsed_time(b) * 1e3)
t0 = time.perf_counter()
for _ in range(1000):
    word[:1].zero_()
host = (time.perf_counter() - t0) / 1000 * 1e6
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 1000 * 1e6
print(f"floor (torch fill of one element): event {best:6.2f} us, host {host:5.2f} us/call, "
      f"back-to-back {total:6.2f} us/launch", flush=True)
for n in sizes:
    lda = 24000
    A = torch.rand(n * lda, dtype=torch.float32, device="cuda").mul_(2 / n ** 0.5).sub_(1 / n ** 0.5)
    x = torch.rand(n, dtype=torch.float32, device="cuda")
    line = f"n={n:5d}"
    for name, lib in zip(names, libs):
        def call():
            assert lib.accblas_acc_trsv(2, 1, 0, 1, n, A.data_ptr(), lda, x.data_ptr(), 1,
                                        helper.data_ptr(), None) == 0
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        # event time of single launches (the CLI's protocol)
        best = 1e9
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3)
        # host time per call, queue kept busy
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            call()
        host = (time.perf_counter() - t0) / 1000 * 1e6
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / 1000 * 1e6
        line += f" | {name}: event {best:6.2f} us, host {host:5.2f} us/call, back-to-back {total:6.2f} us/solve"
    print(line, flush=True)
