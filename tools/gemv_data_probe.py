#!/usr/bin/env python3
"""Developer probe (GPU box): why bench.py's headline GEMV (32768^2, Acc<fp64,
fp32>) takes 4-5 % longer than the tuning tools' - 50 launches back to back per
line, us per launch. It is the VALUES: a matrix rounded from fp64 U(-1,1)
draws (every mantissa bit random, what the reference's generator and bench.py
produce) streams slower than torch.rand()'s 24-bit draws scaled to (-1,1)
(trailing mantissa bits mostly zero), wherever it lies and whatever beta,
stream or res are."""
import ctypes, os, sys, statistics, torch
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", "lib", "libaccblas_hip.so"))
i64, vp, dbl = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
lib.accblas_acc_gemv.restype = ctypes.c_int
lib.accblas_acc_gemv.argtypes = [ctypes.c_int, ctypes.c_int, i64, i64, dbl, vp, i64, vp, i64, dbl, vp, i64, vp]
n = 32768
def batch(A, x, y, beta, stream):
    sp = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5):
            lib.accblas_acc_gemv(2, 1, n, n, 1.0, A.data_ptr(), n, x.data_ptr(), 1, beta, y.data_ptr(), 1, sp)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            lib.accblas_acc_gemv(2, 1, n, n, 1.0, A.data_ptr(), n, x.data_ptr(), 1, beta, y.data_ptr(), 1, sp)
        b.record()
        b.synchronize()
    return a.elapsed_time(b) * 1e3 / 50
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.zeros(n, device="cuda")
A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2).sub_(1)
print("fresh float matrix, beta 0, default stream:", round(batch(A, x, y, 0.0, None), 1))
print("                    beta 1, default stream:", round(batch(A, x, y, 1.0, None), 1))
s = torch.cuda.Stream()
print("                    beta 1, side stream   :", round(batch(A, x, y, 1.0, s), 1))
del A
torch.cuda.empty_cache()
A64 = torch.rand(n * n, dtype=torch.float64, device="cuda").mul_(2).sub_(1)
A = A64.float()
print("float matrix next to its fp64 master      :", round(batch(A, x, y, 1.0, None), 1))
del A64
torch.cuda.empty_cache()
print("after the master is freed                 :", round(batch(A, x, y, 1.0, None), 1))
B = A.clone()
print("a clone of it                             :", round(batch(B, x, y, 1.0, None), 1))
y2 = (torch.rand(n, device="cuda") * 2 - 1)
print("res = random values instead of zeros      :", round(batch(B, x, y2, 1.0, None), 1))
