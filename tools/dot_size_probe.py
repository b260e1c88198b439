import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "accessor-blas_amd", "python"))
import accblas
F32, F64 = accblas.F32, accblas.F64
h = accblas.Handle()
def t(n, ar, st):
    x = torch.rand(n, dtype=accblas.torch_dtype(st), device="cuda"); y = torch.rand(n, dtype=accblas.torch_dtype(st), device="cuda")
    res = torch.zeros(1, dtype=accblas.torch_dtype(ar), device="cuda")
    fn = lambda: accblas.acc_dot(h, ar, st, ar, n, x, 1, y, 1, res)
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(10):
        a, b = accblas.Event(), accblas.Event(); a.record(); fn(); b.record(); b.synchronize(); best = min(best, a.elapsed_ms(b))
    print(f"n={n:>11d} st={accblas.DTYPE_NAMES[st]}: {best*1e3:8.1f} us {2*accblas.DTYPE_BYTES[st]*n/best/1e6:8.1f} GB/s", flush=True)
for n in (201000000, 67108864, 67000000, 33554432, 33000000, 16777216, 17000000, 8388608, 9000000):
    t(n, F64, F32)
for n in (1<<27, 101000000):
    t(n, F64, F64)
