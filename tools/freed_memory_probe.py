#!/usr/bin/env python3
"""Developer probe (GPU box): launches that run after device memory has been
handed back to the driver (torch.cuda.empty_cache() -> hipFree of large
blocks) stream 2-5 % slower for seconds, on buffers that were never touched by
the free - until the memory is allocated again. The headline GEMV (32768^2,
Acc<fp64, fp32>), 50 launches back to back per line, us per launch.
bench.py used to free its 8.6 GB fp64 master right before its timed loop."""
import ctypes, os, sys, time, torch
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", "lib", "libaccblas_hip.so"))
i64, vp, dbl = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
lib.accblas_acc_gemv.restype = ctypes.c_int
lib.accblas_acc_gemv.argtypes = [ctypes.c_int, ctypes.c_int, i64, i64, dbl, vp, i64, vp, i64, dbl, vp, i64, vp]
n = 32768
def batch(A, x, y):
    for _ in range(5):
        lib.accblas_acc_gemv(2, 1, n, n, 1.0, A.data_ptr(), n, x.data_ptr(), 1, 1.0, y.data_ptr(), 1, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lib.accblas_acc_gemv(2, 1, n, n, 1.0, A.data_ptr(), n, x.data_ptr(), 1, 1.0, y.data_ptr(), 1, None)
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / 50
def show(label, A, x, y):
    print(f"{label:60s} {batch(A, x, y):7.1f} us", flush=True)
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.zeros(n, device="cuda")
A = torch.rand(n * n, dtype=torch.float32, device="cuda")
show("fresh A", A, x, y)
show("again", A, x, y)
torch.cuda.empty_cache()
show("after empty_cache() with nothing cached", A, x, y)
D = torch.empty(n * n, dtype=torch.float64, device="cuda")
show("8.6 GB more allocated (untouched)", A, x, y)
D.zero_()
show("... and written once", A, x, y)
del D
show("tensor deleted, block kept in torch's cache", A, x, y)
torch.cuda.empty_cache()
show("empty_cache(): the 8.6 GB go back to the driver", A, x, y)
show("again", A, x, y)
time.sleep(3)
show("3 s later", A, x, y)
D = torch.empty(n * n, dtype=torch.float64, device="cuda"); D.zero_()
show("8.6 GB allocated and written again", A, x, y)
B = A.clone()
show("a clone of A", B, x, y)
del D; torch.cuda.empty_cache()
show("clone, after freeing the 8.6 GB again", B, x, y)
show("A itself", A, x, y)
