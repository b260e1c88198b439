import os, sys, torch
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas
F64 = accblas.F64
lda = 24000
A = (torch.rand(lda * lda, dtype=torch.float64, device="cuda") * 2 - 1) / lda ** 0.5
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
def t(n, fn, reps=8):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = accblas.Event(), accblas.Event(); a.record(); fn(); b.record(); b.synchronize()
        best = min(best, a.elapsed_ms(b) * 1e3)
    return best
for n in (12000, 12016, 12096, 12100, 12104, 12160, 12200, 12288, 12400):
    x0 = torch.rand(n, dtype=torch.float64, device="cuda"); x = x0.clone()
    out = []
    for wide, pf in ((0, 0), (2, 0), (2, 2), (1, 1)):
        accblas.set_option("trsv_wide", wide); accblas.set_option("trsv_prefetch", pf)
        def solve():
            x.copy_(x0); accblas.acc_trsv(F64, F64, accblas.LOWER, accblas.UNIT, n, A, lda, x, 1, helper)
        out.append(f"wide={wide} pf={pf}: {t(n, solve):7.1f}")
    print(n, "n%16 =", n % 16, " ".join(out), flush=True)
