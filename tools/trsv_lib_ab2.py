#!/usr/bin/env python3
"""Developer A/B (GPU box): acc_trsv of SEVERAL builds of libaccblas_hip.so in
ONE process on the same buffers, solve by solve in rotating order (boxes, and
processes on one box, differ by up to 4 % at times - more than most changes).
Acc<fp64, st>, lower (UPLO=upper in the environment: upper) / unit; min and median of `reps` event-timed solves per
build and size (x restored by a device copy inside the timed region, the
copy's own time subtracted).
usage: trsv_lib_ab2.py st reps lib_dir[,lib_dir...] [sizes...]   (lib_dir under accessor-blas_amd/;
lib_dir:option=value sets an accblas option in that build - options are per loaded file, so two
settings of one build need two copies of its directory)"""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ST = {"f16": (0, torch.float16), "f32": (1, torch.float32), "f64": (2, torch.float64)}
F64, UNIT = 2, 1
AR = {"f32": 1, "f64": 2}[os.environ.get("AR", "f64")]  # arithmetic type (AR=f32 in the environment)
UPLO = 0 if os.environ.get("UPLO", "lower") == "upper" else 1  # accblas_uplo

st_name, reps = sys.argv[1], int(sys.argv[2])
names = sys.argv[3].split(",")
sizes = [int(a) for a in sys.argv[4:]] or [8192, 24000, 32768, 40000, 49152, 65536]
st, st_torch = ST[st_name]
i64, vp = ctypes.c_int64, ctypes.c_void_p
libs = []
for name in names:
    directory, _, setting = name.partition(":")
    lib = ctypes.CDLL(os.path.join(REPO, "accessor-blas_amd", directory, "libaccblas_hip.so"))
    for one in filter(None, setting.split("+")):  # option=value[+option=value...]
        option, _, value = one.partition("=")
        lib.accblas_set_option.restype = ctypes.c_int
        lib.accblas_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        assert lib.accblas_set_option(option.encode(), int(value)) == 0, one
    lib.accblas_acc_trsv.restype = ctypes.c_int
    lib.accblas_acc_trsv.argtypes = [ctypes.c_int] * 4 + [i64, vp, i64, vp, i64, vp, vp]
    libs.append(lib)
helper = torch.zeros(2, dtype=torch.int32, device="cuda")
print(f"st={st_name} ar={os.environ.get('AR', 'f64')} {'upper' if UPLO == 0 else 'lower'} reps={reps}; per build: min / median us", flush=True)
for n in sizes:
    scale = 1.0 / n ** 0.5
    # (LDA in the environment: row stride, as the CLIs' sweeps have it; LDA_PLUS: lda = n + that, e.g. an odd stride)
    lda = max(n, int(os.environ.get("LDA", "0"))) + int(os.environ.get("LDA_PLUS", "0"))
    A = torch.rand(n * lda, dtype=torch.float32, device="cuda").mul_(2 * scale).sub_(scale).to(st_torch)
    x0 = torch.rand(n, dtype=torch.float32, device="cuda").mul_(2).sub_(1).to(st_torch)
    x = x0.clone()

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e3

    def solve(lib):
        x.copy_(x0)
        rc = lib.accblas_acc_trsv(AR, st, UPLO, UNIT, n, A.data_ptr(), lda, x.data_ptr(), 1,
                                  helper.data_ptr(), None)
        assert rc == 0, rc
    t_copy = min(timed(lambda: x.copy_(x0)) for _ in range(8))
    times = [[] for _ in libs]
    # (bits are compared per build: builds with different row-block splits sum
    # in different groupings)
    first = [None] * len(libs)
    for rep in range(-1, reps):
        for k in range(len(libs)):
            i = (k + rep) % len(libs)
            t = timed(lambda: solve(libs[i])) - t_copy
            if rep >= 0:
                times[i].append(t)
            assert helper[0].item() & 0xffffffff == 0xffffffff
            if first[i] is None:
                first[i] = x.clone()
            assert torch.equal(first[i], x), f"{names[i]} is not repeatable"
    # SAME_BITS=1 in the environment: every build must give the first one's bits (builds that differ in
    # waiting and scheduling only)
    same = all(torch.equal(first[0], other) for other in first[1:])
    assert same or os.environ.get("SAME_BITS", "0") == "0", "builds differ in their results"
    print(f"n={n:6d} " + "  ".join(f"{names[i]}: {min(t):7.1f} / {statistics.median(t):7.1f}"
                                   for i, t in enumerate(times)) +
          ("" if len(libs) < 2 else ("   [same bits]" if same else "   [bits differ]")), flush=True)
    del A
