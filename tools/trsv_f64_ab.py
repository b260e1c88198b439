import sys, os, torch
sys.path.insert(0, "/root/repo/accessor-blas_amd/python")
import accblas
F64=accblas.F64
def t(fn,reps=4):
    fn(); torch.cuda.synchronize(); best=1e30
    for _ in range(reps):
        a,b=accblas.Event(),accblas.Event(); a.record(); fn(); b.record(); b.synchronize(); best=min(best,a.elapsed_ms(b)*1e3)
    return best
for n in (16384, 24000, 32768, 49152, 65536):
    s=1/n**0.5
    A=(torch.rand(n*n,dtype=torch.float32,device="cuda").mul_(2*s).sub_(s)).double()
    x0=torch.rand(n,dtype=torch.float64,device="cuda")*2-1; x=x0.clone(); h=torch.zeros(2,dtype=torch.int32,device="cuda")
    r=[]
    for w in (0,2,0,2):
        accblas.set_option("trsv_wide",w)
        def f():
            x.copy_(x0); accblas.acc_trsv(F64,F64,accblas.LOWER,accblas.UNIT,n,A,n,x,1,h)
        r.append(t(f))
    print(f"n={n} fp64/fp64 rows64 {min(r[0],r[2]):.1f} rows128 {min(r[1],r[3]):.1f} ratio {min(r[1],r[3])/min(r[0],r[2]):.3f} helper={h[0].item()&0xffffffff:08x}",flush=True)
    del A
