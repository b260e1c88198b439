"""Kernel parity on the REFERENCE'S OWN INPUTS (north_star: "match the reference
CUDA kernels' outputs on the same random inputs").

The other parity tests draw from numpy; here the operands are exactly what the
reference's benchmarks build before their first launch:
  * GEMV (`/root/reference/cuda/gemv_benchmark.cu:81-87`, `gemv_memory.cuh:39-84`):
    `std::default_random_engine(42)`, U(-1, 1) doubles, drawn in the order
    matrix (max_size x max_size, row-major, 24500 by default) -> x -> res; the
    storage-type holders are `static_cast` copies; a problem of n rows is the
    top-left n x n block with row stride max_size (`gemv_benchmark.cu:215`),
    alpha = beta = 1 (`:25-28`);
  * TRSV (`cuda/trsv_benchmark.cu:26-27,78-92`, `trsv_memory.cuh:110-169`): the
    same engine, matrix -> x, then the matrix is LU-factorised IN PLACE on the
    device (its row-major buffer read column-major, pivots dropped), `static_cast`
    copies, upper / unit solves of the top-left blocks.
Each of the reference's five hand-written variants (plain fp64, plain fp32,
Acc<fp64,fp64>, Acc<fp64,fp32>, Acc<fp32,fp32>) goes through the C ABI and is
compared with the order-faithful emulation of the CUDA kernel (`ref_*`) AND
with the long-double oracle (`exact_*`), both fed the same buffers. There are
no reference OUTPUTS to compare with (the repository ships none, and its CUDA
path cannot run here), so this pins inputs and summation semantics, not bits of
an NVIDIA run: tolerances as in tests/test_gpu_parity.py.
"""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as oracle
import trsv_margins
from oracle_lib import F32, F64
from test_gpu_parity import out_tolerance

pytestmark = pytest.mark.gpu

GEMV_MAX_SIZE = 24500          # default of gemv_benchmark (`gemv_benchmark.cu:38`)
GEMV_SIZES = (100, 1300, 4100)
TRSV_MAX_SIZE = 2000           # `trsv_benchmark --size=2000`
TRSV_SIZES = (100, 700, 2000)
# (name, arithmetic, storage, plain pointers?) in the reference's column order
VARIANTS = [("fp64", F64, F64, True), ("fp32", F32, F32, True),
            ("Acc<fp64, fp64>", F64, F64, False), ("Acc<fp64, fp32>", F64, F32, False),
            ("Acc<fp32, fp32>", F32, F32, False)]


@pytest.fixture(scope="module")
def gpu():
    import torch
    import accblas
    accblas.load()
    assert torch.cuda.is_available()
    return torch, accblas


@pytest.fixture(scope="module")
def gemv_inputs():
    """The leading rows of the reference's 24500 x 24500 matrix and its x / res,
    with the engine advanced past the rows that are not kept."""
    rows_kept = max(GEMV_SIZES)
    engine = oracle.Engine(42)
    A = engine.gen_mtx(rows_kept, GEMV_MAX_SIZE)            # rows 0 .. rows_kept-1
    engine.discard((GEMV_MAX_SIZE - rows_kept) * GEMV_MAX_SIZE)
    x = engine.gen_mtx(GEMV_MAX_SIZE, 1)
    res = engine.gen_mtx(GEMV_MAX_SIZE, 1)
    return {F64: (A, x, res),
            F32: tuple(oracle.convert(v, F32) for v in (A, x, res))}


def test_generator_stream_is_the_reference_one(gemv_inputs):
    """First draws of minstd_rand0(42) through uniform_real_distribution<double>
    (two engine steps per value, libstdc++ generate_canonical), and the x / res
    vectors sit where 24500^2 matrix draws put them."""
    A, x, res = gemv_inputs[F64]
    engine = oracle.Engine(42)
    head = engine.gen_mtx(1, 4)
    assert np.array_equal(A[:4], head)
    # by hand: minstd_rand0 x <- 16807 x mod (2^31 - 1); canonical =
    # ((a - 1) + (b - 1) * R) / R^2 with R = 2^31 - 2
    state, R = 42, 2147483646
    draws = []
    for _ in range(4):
        state = state * 16807 % 2147483647
        a = state
        state = state * 16807 % 2147483647
        b = state
        draws.append(((a - 1) + (b - 1) * float(R)) / (float(R) * float(R)) * 2.0 - 1.0)
    assert np.allclose(head, draws, rtol=0, atol=1e-15)
    again = oracle.Engine(42)
    again.discard(GEMV_MAX_SIZE * GEMV_MAX_SIZE)
    assert np.array_equal(again.gen_mtx(GEMV_MAX_SIZE, 1), x)
    assert np.array_equal(again.gen_mtx(GEMV_MAX_SIZE, 1), res)
    assert -1.0 <= A.min() and A.max() < 1.0 and abs(A.mean()) < 1e-3


@pytest.mark.parametrize("n", GEMV_SIZES)
def test_gemv_variants_on_the_reference_inputs(gpu, gemv_inputs, n):
    torch, accblas = gpu
    lda = GEMV_MAX_SIZE
    results = {}
    for name, ar, st, plain in VARIANTS:
        A, x, res = gemv_inputs[st]
        dA = torch.from_numpy(A[: n * lda]).cuda()
        dx = torch.from_numpy(x[:n].copy()).cuda()
        dres = torch.from_numpy(res[:n].copy()).cuda()
        if plain:
            accblas.gemv(ar, n, n, 1.0, dA, lda, dx, 1, 1.0, dres, 1)
        else:
            accblas.acc_gemv(ar, st, n, n, 1.0, dA, lda, dx, 1, 1.0, dres, 1)
        torch.cuda.synchronize()
        got = dres.cpu().numpy()
        results[name] = got
        tol = out_tolerance(ar, st)
        emulated = oracle.ref_gemv(ar, n, n, 1.0, A, lda, x, 1, 1.0, res[:n], 1)
        exact = oracle.exact_gemv(ar, n, n, 1.0, A, lda, x, 1, 1.0, res[:n], 1)
        for label, want in (("ref_gemv", emulated), ("exact_gemv", exact)):
            err = oracle.rel_error(want.astype(np.float64), got.astype(np.float64))
            assert err <= tol, (name, n, label, err)
        del dA
    # the reference's premise: the accessor costs no accuracy (same bits as the
    # plain kernel of the same type)
    assert np.array_equal(results["fp64"], results["Acc<fp64, fp64>"])
    assert np.array_equal(results["fp32"], results["Acc<fp32, fp32>"])
    # and its `--error` column (error against the fp64 kernel on the unrounded
    # data, `gemv_benchmark.cu:219-232`): fp64 arithmetic on fp32 storage beats
    # the fp32 kernel
    ref64 = results["fp64"]
    err_mixed = oracle.rel_error(ref64, results["Acc<fp64, fp32>"].astype(np.float64))
    err_fp32 = oracle.rel_error(ref64, results["fp32"].astype(np.float64))
    assert err_mixed < err_fp32
    assert 1e-8 < err_mixed < 8e-8, err_mixed


@pytest.fixture(scope="module")
def trsv_inputs(gpu):
    """gen A -> gen x -> LU on the device in place (rocSOLVER standing in for
    cusolverDnDgetrf), then the `static_cast` copies."""
    torch, accblas = gpu
    n = TRSV_MAX_SIZE
    engine = oracle.Engine(42)
    A = engine.gen_mtx(n, n)
    x = engine.gen_mtx(n, 1)
    lib = ctypes.CDLL(os.path.join(accblas.PACKAGE_ROOT, "lib", "libaccblas_vendor.so"))
    lib.accblas_vendor_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.accblas_vendor_destroy.argtypes = [ctypes.c_void_p]
    lib.accblas_vendor_getrf.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
    handle = ctypes.c_void_p()
    assert lib.accblas_vendor_create(ctypes.byref(handle)) == 0
    dA = torch.from_numpy(A).cuda()
    info = ctypes.c_int(-1)
    assert lib.accblas_vendor_getrf(handle, n, dA.data_ptr(), n, ctypes.byref(info)) == 0
    torch.cuda.synchronize()
    lib.accblas_vendor_destroy(handle)
    assert info.value == 0
    lu = dA.cpu().numpy()
    # what the benchmark relies on: the row-major upper triangle is the unit
    # factor of the column-major LU, bounded by 1 (partial pivoting)
    assert np.abs(np.triu(lu.reshape(n, n), 1)).max() <= 1.0 + 1e-15
    return {F64: (lu, x), F32: (oracle.convert(lu, F32), oracle.convert(x, F32))}


@pytest.mark.parametrize("n", TRSV_SIZES)
def test_trsv_variants_on_the_reference_inputs(gpu, trsv_inputs, n):
    torch, accblas = gpu
    lda = TRSV_MAX_SIZE
    results = {}
    for name, ar, st, plain in VARIANTS:
        A, x = trsv_inputs[st]
        dA = torch.from_numpy(A).cuda()
        dx = torch.from_numpy(x[:n].copy()).cuda()
        helper = torch.zeros(2, dtype=torch.int32, device="cuda")
        if plain:
            accblas.trsv(ar, accblas.UPPER, accblas.UNIT, n, dA, lda, dx, 1, helper)
        else:
            accblas.acc_trsv(ar, st, accblas.UPPER, accblas.UNIT, n, dA, lda, dx, 1, helper)
        torch.cuda.synchronize()
        assert not accblas.trsv_status(helper)
        got = dx.cpu().numpy()
        results[name] = got
        # x is stored (and re-read by later blocks) in the storage type: the
        # chain amplifies rounding by the factor's condition (tests/test_gpu_parity.py)
        tol = 100 * out_tolerance(ar, st)
        emulated = oracle.ref_trsv(ar, True, True, n, A, lda, x[:n], 1)
        exact = oracle.exact_trsv(ar, True, True, n, A, lda, x[:n], 1)
        for label, want in (("ref_trsv", emulated), ("exact_trsv", exact)):
            err = oracle.rel_error(want.astype(np.float64), got.astype(np.float64))
            trsv_margins.check(trsv_margins.key("reference_inputs_" + ("plain" if plain else "acc"),
                                                ar, st, True, label), err, tol)
    assert np.array_equal(results["fp64"], results["Acc<fp64, fp64>"])
    assert np.array_equal(results["fp32"], results["Acc<fp32, fp32>"])
    ref64 = results["fp64"]
    err_mixed = oracle.rel_error(ref64, results["Acc<fp64, fp32>"].astype(np.float64))
    err_fp32 = oracle.rel_error(ref64, results["fp32"].astype(np.float64))
    assert err_mixed <= err_fp32
