#!/usr/bin/env python3
"""Developer probe (GPU box): the run-to-run spread of acc_trsv. `reps`
event-timed solves per case in one process (x restored by a device copy in
front of the timed region, NOT inside it), every time kept: quantiles, the
ratio mean / min, max / min, and the sequence itself (is a slow solve the
first one, every k-th one, or at random?).
usage: trsv_spread.py reps case[,case...]   case = st:ar:uplo:n[:option=value+...]
  e.g. f32:f64:lower:65536  f32:f64:upper:24000:trsv_prefetch=0
LIB=lib_dir in the environment: another build under accessor-blas_amd/."""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ST = {"f16": (0, torch.float16), "f32": (1, torch.float32), "f64": (2, torch.float64)}
AR = {"f32": 1, "f64": 2}
UNIT = 1
DEFAULTS = {"trsv_wide_split_from": 0, "trsv_wide_persistent": 0}  # (every other option: 1)

reps = int(sys.argv[1])
cases = sys.argv[2].split(",")
i64, vp = ctypes.c_int64, ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", os.environ.get("LIB", "lib"),
                               "libaccblas_hip.so"))
lib.accblas_set_option.restype = ctypes.c_int
lib.accblas_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
lib.accblas_acc_trsv.restype = ctypes.c_int
lib.accblas_acc_trsv.argtypes = [ctypes.c_int] * 4 + [i64, vp, i64, vp, i64, vp, vp]
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
for case in cases:
    fields = case.split(":")
    st_name, ar_name, uplo_name, n = fields[0], fields[1], fields[2], int(fields[3])
    settings = fields[4].split("+") if len(fields) > 4 else []
    for one in settings:
        option, _, value = one.partition("=")
        assert lib.accblas_set_option(option.encode(), int(value)) == 0, one
    st, st_torch = ST[st_name]
    uplo = 0 if uplo_name == "upper" else 1
    scale = 1.0 / n ** 0.5
    A = torch.rand(n * n, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale).to(st_torch)
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(st_torch)
    x = x0.clone()
    times = []
    for rep in range(-2, reps):
        x.copy_(x0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.accblas_acc_trsv(AR[ar_name], st, uplo, UNIT, n, A.data_ptr(), n, x.data_ptr(), 1,
                                  helper.data_ptr(), None)
        b.record()
        b.synchronize()
        assert rc == 0, rc
        assert helper[0].item() & 0xffffffff == 0xffffffff
        if rep >= 0:
            times.append(a.elapsed_time(b) * 1e3)
    ordered = sorted(times)
    q = lambda f: ordered[min(len(ordered) - 1, int(f * len(ordered)))]
    mean = statistics.fmean(times)
    print(f"{case}: min {ordered[0]:.1f} p10 {q(.1):.1f} p50 {q(.5):.1f} p90 {q(.9):.1f} "
          f"max {ordered[-1]:.1f} mean {mean:.1f} us; mean/min {mean / ordered[0]:.3f} "
          f"max/min {ordered[-1] / ordered[0]:.3f}", flush=True)
    print("   seq: " + " ".join(f"{t:.0f}" for t in times), flush=True)
    for one in settings:  # back to the library's defaults
        option = one.partition("=")[0]
        lib.accblas_set_option(option.encode(), DEFAULTS.get(option, 1))
    del A
