"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle.

Run with `pytest -m gpu` on an MI355X. Metric and tolerances follow SURVEY.md
§8d D4, check 1 (kernel parity on the SAME storage-rounded inputs, so only the
summation order differs):
  GEMV / TRSV: sum|oracle - hip| / sum|oracle|  (the reference's own `--error`
  metric, `/root/reference/cuda/gemv_benchmark.cu:101-124`)
      Ar=f64, f64 output ........ <= 1e-13
      Ar=f64, f32 output ........ <= 1e-7   (rare 1-ulp(f32) flips)
      Ar=f32, f32/f64 output .... <= 5e-6
      any f16 output ............ <= 1e-3
  DOT: |oracle - hip| <= 64 * eps(Ar) * Q, Q = sqrt(sum (x_i y_i)^2)
  plain vs Acc<T,T>: exactly 0 (bit for bit).
"""
import numpy as np
import pytest

import oracle_lib as oracle
import trsv_margins
from oracle_lib import F16, F32, F64

pytestmark = pytest.mark.gpu

PAIRS = [(F64, F64), (F64, F32), (F32, F32), (F64, F16), (F32, F16), (F32, F64)]
EPS = {F32: 2.0 ** -24, F64: 2.0 ** -53}


def out_tolerance(ar, st):
    if st == F16:
        return 1e-3
    if ar == F64 and st == F64:
        return 1e-13
    if ar == F64 and st == F32:
        return 1e-7
    return 5e-6


@pytest.fixture(scope="module")
def gpu():
    import torch
    import accblas
    accblas.load()
    assert torch.cuda.is_available()
    return torch, accblas


def to_dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def make_inputs(seed, count, st):
    rng = np.random.default_rng(seed)
    return oracle.convert(rng.uniform(-1.0, 1.0, count), st)


# --------------------------------------------------------------------------- GEMV
GEMV_SHAPES = [
    # rows, cols, lda, offset (elements the base pointers are shifted by)
    (1, 1, 1, 0),
    (37, 53, 64, 0),      # ragged columns, vector path
    (100, 100, 24500 // 4 * 4, 0),
    (513, 1000, 1024, 0),
    (64, 4096, 4096, 0),
    (2048, 2048, 2048, 0),
    # rows that are not 16-byte aligned: 16-byte loads all the same since round 4
    # (until then: 8-byte loads / the element-wise kernel, option "gemv_vector_rows" = 0)
    (130, 1027, 1031, 0),  # lda odd
    (200, 512, 520, 1),    # misaligned base
    (37, 13001, 13003, 1),  # both, long rows: every row on its own 128-byte aligned grid
    (9, 26001, 26003, 3),
    (3, 70000, 70000, 0),  # few long rows
    (37, 13000, 13004, 0),  # pitch not a multiple of 128 bytes: per-row aligned grids
    (9, 26001, 26004, 0),
    # the small end of the reference's sweep (top-left blocks of a 24500-wide
    # matrix): short rows, one row per workgroup
    (1300, 1300, 24500, 0),
    (1900, 1900, 24500, 0),
    (2200, 2200, 24500, 0),
    (5000, 1500, 1504, 0),  # short rows, many of them: the two-row shape
    # rows that count as short for 8-byte (up to 40 KiB) / 4-byte (28 KiB) storage only
    (600, 4100, 4104, 0),
    (500, 6500, 6504, 0),
]


@pytest.mark.parametrize("ar,st", PAIRS)
@pytest.mark.parametrize("rows,cols,lda,offset", GEMV_SHAPES)
def test_acc_gemv_matches_oracle(gpu, ar, st, rows, cols, lda, offset):
    torch, accblas = gpu
    A = make_inputs(1, rows * lda + offset, st)
    x = make_inputs(2, cols + offset, st)
    res = make_inputs(3, rows, st)
    alpha, beta = 1.0, 1.0
    want = oracle.exact_gemv(ar, rows, cols, alpha, A[offset:], lda, x[offset:], 1, beta,
                             res, 1)
    dA, dx, dres = to_dev(torch, A), to_dev(torch, x), to_dev(torch, res)
    accblas.acc_gemv(ar, st, rows, cols, alpha, dA[offset:], lda, dx[offset:], 1, beta,
                     dres, 1)
    torch.cuda.synchronize()
    got = dres.cpu().numpy()
    err = oracle.rel_error(want.astype(np.float64), got.astype(np.float64))
    assert err <= out_tolerance(ar, st), (err, rows, cols)
    # the order-faithful emulation of the CUDA kernel must agree to the same bar
    emulated = oracle.ref_gemv(ar, rows, cols, alpha, A[offset:], lda, x[offset:], 1,
                               beta, res, 1)
    err2 = oracle.rel_error(emulated.astype(np.float64), got.astype(np.float64))
    assert err2 <= out_tolerance(ar, st), (err2, rows, cols)


@pytest.mark.parametrize("vt", [F32, F64])
@pytest.mark.parametrize("rows,cols,lda,offset", GEMV_SHAPES)
def test_plain_gemv_equals_accessor_bitwise(gpu, vt, rows, cols, lda, offset):
    """BASELINE config 2: plain-pointer kernel == Acc<T,T> kernel, bit for bit."""
    torch, accblas = gpu
    A = to_dev(torch, make_inputs(4, rows * lda + offset, vt))
    x = to_dev(torch, make_inputs(5, cols + offset, vt))
    res0 = make_inputs(6, rows, vt)
    r_plain, r_acc = to_dev(torch, res0), to_dev(torch, res0)
    accblas.gemv(vt, rows, cols, 1.0, A[offset:], lda, x[offset:], 1, 1.0, r_plain, 1)
    accblas.acc_gemv(vt, vt, rows, cols, 1.0, A[offset:], lda, x[offset:], 1, 1.0, r_acc,
                     1)
    torch.cuda.synchronize()
    assert torch.equal(r_plain, r_acc)
    assert not torch.equal(r_plain, to_dev(torch, res0)) or cols == 0


@pytest.mark.parametrize("cols", [2000, 4500])  # 4500: short for fp64 and fp32 storage only by the longer limits
@pytest.mark.parametrize("ar,st", [(F64, F32), (F64, F64), (F32, F16)])
def test_gemv_row_blocks_of_short_rows_keep_their_bits(gpu, ar, st, cols):
    """Short rows pick one of two workgroup shapes by the NUMBER of rows; a
    row's summation order is the same in both, so a row block (what one GPU of
    a sharded GEMV computes) has the bits of the same rows inside the taller
    product, on either side of the row-count rule."""
    torch, accblas = gpu
    rows, lda = 6000, cols + 8
    A = to_dev(torch, make_inputs(41, rows * lda, st))
    x = to_dev(torch, make_inputs(42, cols, st))
    res0 = make_inputs(43, rows, st)
    full = to_dev(torch, res0)
    accblas.acc_gemv(ar, st, rows, cols, 1.0, A, lda, x, 1, 1.0, full, 1)
    for lo, hi in [(0, 1000), (1000, 1001), (1024, 5500), (5000, 6000)]:
        part = to_dev(torch, res0)
        accblas.acc_gemv(ar, st, hi - lo, cols, 1.0, A[lo * lda:], lda, x, 1, 1.0,
                         part[lo:], 1)
        torch.cuda.synchronize()
        assert torch.equal(part[lo:hi], full[lo:hi]), (lo, hi)


def test_gemv_beta_zero_does_not_read_res(gpu):
    """beta == 0 must skip the read (`cuda/gemv_kernels.cuh:58-59,107-108`)."""
    torch, accblas = gpu
    rows, cols = 300, 700
    A = make_inputs(7, rows * cols, F32)
    x = make_inputs(8, cols, F32)
    res = np.full(rows, np.nan, dtype=np.float32)
    want = oracle.exact_gemv(F64, rows, cols, 0.5, A, cols, x, 1, 0.0, res, 1)
    dres = to_dev(torch, res)
    accblas.acc_gemv(F64, F32, rows, cols, 0.5, to_dev(torch, A), cols, to_dev(torch, x),
                     1, 0.0, dres, 1)
    torch.cuda.synchronize()
    got = dres.cpu().numpy()
    assert np.all(np.isfinite(got))
    assert oracle.rel_error(want.astype(np.float64), got.astype(np.float64)) <= 1e-7


def test_gemv_strided_vectors(gpu):
    """x and res with increments != 1 (`x[col * x_info.stride]`, gemv_kernels.cuh:51)."""
    torch, accblas = gpu
    rows, cols, incx, incres = 257, 333, 3, 2
    A = make_inputs(9, rows * cols, F32)
    x = make_inputs(10, cols * incx, F32)
    res = make_inputs(11, rows * incres, F32)
    want = oracle.exact_gemv(F64, rows, cols, -1.5, A, cols, x, incx, 0.25, res, incres)
    dres = to_dev(torch, res)
    accblas.acc_gemv(F64, F32, rows, cols, -1.5, to_dev(torch, A), cols, to_dev(torch, x),
                     incx, 0.25, dres, incres)
    torch.cuda.synchronize()
    got = dres.cpu().numpy()
    # untouched gaps stay untouched
    mask = np.ones(rows * incres, dtype=bool)
    mask[::incres] = False
    assert np.array_equal(got[mask], res[mask])
    err = oracle.rel_error(want[::incres].astype(np.float64),
                           got[::incres].astype(np.float64))
    assert err <= 1e-7


@pytest.mark.parametrize("st,tol", [(F16, 1e-3), (F32, 1e-7), (F64, 1e-13)])
@pytest.mark.parametrize("incx,x_offset", [(3, 0), (1, 1), (2, 1)])
def test_gemv_vector_matrix_loads_with_strided_or_unaligned_x(gpu, st, tol, incx, x_offset):
    """Aligned matrix, x strided and/or off the 16-byte grid: 16-byte matrix
    loads + element-wise x (`gemv_vec<.., VecX = false>`); ragged cols and rows."""
    torch, accblas = gpu
    rows, cols, lda = 203, 4099, 4112
    A = make_inputs(31, rows * lda, st)
    x = make_inputs(32, cols * incx + x_offset, st)
    res = make_inputs(33, rows, st)
    want = oracle.exact_gemv(F64, rows, cols, 0.75, A, lda, x[x_offset:], incx, -0.5, res, 1)
    dres = to_dev(torch, res)
    accblas.acc_gemv(F64, st, rows, cols, 0.75, to_dev(torch, A), lda,
                     to_dev(torch, x)[x_offset:], incx, -0.5, dres, 1)
    torch.cuda.synchronize()
    got = dres.cpu().numpy()
    assert oracle.rel_error(want.astype(np.float64), got.astype(np.float64)) <= tol


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("ar,st", [(F64, F32), (F64, F16), (F64, F64), (F32, F32)])
@pytest.mark.parametrize("rows,cols,lda", [(203, 4096, 4096), (130, 4096 + 37, 8192),
                                           (64, 2048, 2048)])
def test_gemv_plain_and_time_skewed_sweeps_match_oracle(gpu, mode, ar, st, rows, cols, lda):
    """Option "gemv_skew": 0 = plain sweep, 2 = time-skewed sweep whenever the
    shape allows it (power-of-two pitches use it by default with fp64 arithmetic).
    Both must match the oracle, with ragged rows / columns, and the plain-pointer
    kernel must stay bitwise equal to the accessor kernel in either mode."""
    torch, accblas = gpu
    A = make_inputs(71, rows * lda, st)
    x = make_inputs(72, cols, st)
    res = make_inputs(73, rows, st)
    want = oracle.exact_gemv(ar, rows, cols, 1.25, A, lda, x, 1, -0.5, res, 1)
    accblas.set_option("gemv_skew", mode)
    try:
        dres = to_dev(torch, res)
        accblas.acc_gemv(ar, st, rows, cols, 1.25, to_dev(torch, A), lda, to_dev(torch, x), 1,
                         -0.5, dres, 1)
        if ar == st:
            plain = to_dev(torch, res)
            accblas.gemv(st, rows, cols, 1.25, to_dev(torch, A), lda, to_dev(torch, x), 1,
                         -0.5, plain, 1)
            torch.cuda.synchronize()
            assert torch.equal(plain, dres)
        torch.cuda.synchronize()
    finally:
        accblas.set_option("gemv_skew", 1)
    err = oracle.rel_error(want.astype(np.float64), dres.cpu().numpy().astype(np.float64))
    assert err <= out_tolerance(ar, st), err


def test_gemv_empty_and_errors(gpu):
    torch, accblas = gpu
    dummy = torch.zeros(4, dtype=torch.float32, device="cuda")
    accblas.acc_gemv(F64, F32, 0, 0, 1.0, dummy, 1, dummy, 1, 1.0, dummy, 1)
    accblas.acc_gemv(F64, F32, 0, 5, 1.0, dummy, 5, dummy, 1, 1.0, dummy, 1)
    # rows > 0, cols == 0: res = beta * res
    r = torch.tensor([2.0, 4.0], dtype=torch.float32, device="cuda")
    accblas.acc_gemv(F64, F32, 2, 0, 1.0, dummy, 1, dummy, 1, 0.5, r, 1)
    torch.cuda.synchronize()
    assert r.tolist() == [1.0, 2.0]
    with pytest.raises(accblas.AccblasError) as info:
        accblas.acc_gemv(F64, F32, 2, 4, 1.0, dummy, 3, dummy, 1, 1.0, dummy, 1)
    assert info.value.status == -2
    with pytest.raises(accblas.AccblasError) as info:
        accblas.acc_gemv(F16, F32, 2, 2, 1.0, dummy, 2, dummy, 1, 1.0, dummy, 1)
    assert info.value.status == -1


# ---------------------------------------------------------------------------- DOT
DOT_SIZES = [0, 1, 7, 64, 1000, 4099, 1_000_000, (1 << 22) + 3]


@pytest.mark.parametrize("ar,st", PAIRS)
@pytest.mark.parametrize("n", DOT_SIZES)
@pytest.mark.parametrize("mode", [0, 1])
def test_acc_dot_matches_oracle(gpu, ar, st, n, mode):
    torch, accblas = gpu
    handle = accblas.Handle()
    handle.set_dot_mode(mode)
    x = make_inputs(20, max(n, 1), st)
    y = make_inputs(21, max(n, 1), st)
    want, scale = oracle.exact_dot(ar, n, x, 1, y, 1)
    res = torch.full((1,), -999.0, dtype=accblas.torch_dtype(ar), device="cuda")
    dx, dy = to_dev(torch, x), to_dev(torch, y)
    accblas.acc_dot(handle, ar, st, ar, n, dx, 1, dy, 1, res)
    torch.cuda.synchronize()
    got = float(res.item())
    assert abs(got - want) <= 64 * EPS[ar] * scale + 1e-300, (got, want, scale)
    if mode == 0:
        # deterministic mode: bitwise reproducible
        res2 = torch.full((1,), -1.0, dtype=accblas.torch_dtype(ar), device="cuda")
        accblas.acc_dot(handle, ar, st, ar, n, dx, 1, dy, 1, res2)
        torch.cuda.synchronize()
        assert torch.equal(res, res2)
    handle.close()


@pytest.mark.parametrize("ar,st", PAIRS)
@pytest.mark.parametrize("x_offset,y_offset", [(1, 0), (3, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_acc_dot_of_subvectors_off_the_vector_alignment(gpu, ar, st, x_offset, y_offset, mode):
    """Subvectors that start off a 16-byte boundary (a view into a larger vector,
    `cuda/dot_kernels.cuh:209-240` builds the ranges from whatever pointers it is
    given): the same 16-byte loads since round 4, element-wise until then."""
    torch, accblas = gpu
    handle = accblas.Handle()
    handle.set_dot_mode(mode)
    n = 1_000_003
    x = make_inputs(26, n + x_offset, st)
    y = make_inputs(27, n + y_offset, st)
    want, scale = oracle.exact_dot(ar, n, x[x_offset:], 1, y[y_offset:], 1)
    res = torch.zeros(1, dtype=accblas.torch_dtype(ar), device="cuda")
    accblas.acc_dot(handle, ar, st, ar, n, to_dev(torch, x)[x_offset:], 1,
                    to_dev(torch, y)[y_offset:], 1, res)
    torch.cuda.synchronize()
    assert abs(float(res.item()) - want) <= 64 * EPS[ar] * scale + 1e-300
    handle.close()


@pytest.mark.parametrize("vt", [F32, F64])
def test_plain_dot_equals_accessor_bitwise(gpu, vt):
    torch, accblas = gpu
    handle = accblas.Handle()
    n = 777_777
    dx, dy = to_dev(torch, make_inputs(22, n, vt)), to_dev(torch, make_inputs(23, n, vt))
    r1 = torch.zeros(1, dtype=accblas.torch_dtype(vt), device="cuda")
    r2 = torch.zeros(1, dtype=accblas.torch_dtype(vt), device="cuda")
    accblas.dot(handle, vt, n, dx, 1, dy, 1, r1)
    accblas.acc_dot(handle, vt, vt, vt, n, dx, 1, dy, 1, r2)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2)
    handle.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_acc_dot_result_type_and_strides(gpu, mode):
    """ResType != ArType (`cast_result`, dot_kernels.cuh:245-262) and increments."""
    torch, accblas = gpu
    handle = accblas.Handle()
    handle.set_dot_mode(mode)
    n, incx, incy = 50_001, 2, 3
    x = make_inputs(24, n * incx, F32)
    y = make_inputs(25, n * incy, F32)
    want, scale = oracle.exact_dot(F64, n, x, incx, y, incy)
    res = torch.zeros(1, dtype=torch.float32, device="cuda")
    accblas.acc_dot(handle, F64, F32, F32, n, to_dev(torch, x), incx, to_dev(torch, y),
                    incy, res)
    torch.cuda.synchronize()
    assert float(res.item()) == np.float32(want) or \
        abs(float(res.item()) - want) <= 2.0 ** -23 * abs(want) + 64 * EPS[F64] * scale
    handle.close()


# --------------------------------------------------------------------------- TRSV
KERNELS = {
    # name -> (trsv_wide, trsv_relay, trsv_small) of accblas_set_option
    "rows64": (0, 1, 0),         # 64-row blocks, loader wave by the measured rule
    "rows64_direct": (0, 0, 0),  # every worker polls the granules it needs
    "rows64_relay": (0, 2, 0),   # x through the loader wave and LDS, all types/sizes
    "rows128": (2, 1, 0),        # 128-row blocks wherever the operands allow it
    "default": (1, 1, 1),        # the library's own choice (n <= 128: one workgroup)
}


@pytest.fixture
def trsv_kernel(request, gpu):
    """Selects the TRSV kernel family for one test, restores the defaults after."""
    _, accblas = gpu
    wide, relay, small = KERNELS[request.param]
    accblas.set_option("trsv_wide", wide)
    accblas.set_option("trsv_relay", relay)
    accblas.set_option("trsv_small", small)
    yield request.param
    accblas.set_option("trsv_wide", 1)
    accblas.set_option("trsv_relay", 1)
    accblas.set_option("trsv_small", 1)


_TRIANGULAR_CACHE = {}


def triangular_system(n, lda, upper, seed):
    key = (n, lda, upper, seed)
    if key not in _TRIANGULAR_CACHE:
        if len(_TRIANGULAR_CACHE) > 64:
            _TRIANGULAR_CACHE.clear()
        _TRIANGULAR_CACHE[key] = _triangular_system(n, lda, upper, seed)
    return _TRIANGULAR_CACHE[key]


def _triangular_system(n, lda, upper, seed):
    """Well-conditioned unit-bounded triangular factor as the reference builds it:
    partial-pivot LU of a U(-1,1) matrix (`cuda/trsv_memory.cuh:131-168`), the
    bounded unit factor placed in the requested triangle, the opposite triangle
    filled with junk that must never be read as part of T, and a diagonal in
    [1, 2) for the non-unit case."""
    import scipy.linalg
    rng = np.random.default_rng(seed)
    dense = rng.uniform(-1.0, 1.0, (n, n))
    _, lower, _ = scipy.linalg.lu(dense)
    strict = np.tril(lower, -1)
    diag = rng.uniform(1.0, 2.0, n)
    full = rng.uniform(-1.0, 1.0, (n, lda))
    tri = strict.T if upper else strict
    mask = np.triu(np.ones((n, n), dtype=bool), 1) if upper else \
        np.tril(np.ones((n, n), dtype=bool), -1)
    full[:, :n][mask] = tri[mask]
    full[np.arange(n), np.arange(n)] = diag
    return full.reshape(-1)


TRSV_SIZES = [(1, 1), (31, 40), (64, 64), (65, 100), (100, 24000 // 8), (128, 128),
              (200, 256), (264, 264), (1000, 1000), (2050, 2112),
              # n off the vector length with a row stride the vector paths take:
              # an upper solve then has a vector across its first real column
              (267, 272), (1001, 1008), (2051, 2112),
              # packed matrices whose row stride is off the vector length: 16-byte
              # loads that are not 16-byte aligned (until round 3: the 64-row
              # kernel's element-wise path, option "trsv_vector_rows" = 0)
              (270, 270), (1002, 1002), (1004, 1004)]


@pytest.mark.parametrize("trsv_kernel", list(KERNELS), indirect=True)
@pytest.mark.parametrize("ar,st", PAIRS)
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("n,lda", TRSV_SIZES)
def test_acc_trsv_matches_oracle(gpu, trsv_kernel, ar, st, upper, unit, n, lda):
    """Every kernel family (64-row blocks direct / through the loader wave,
    128-row blocks) x every (arithmetic, storage) pair against both oracle levels."""
    torch, accblas = gpu
    A = oracle.convert(triangular_system(n, lda, upper, 30 + n), st)
    b = make_inputs(31, n, st)
    want = oracle.exact_trsv(ar, upper, unit, n, A, lda, b, 1)
    dA, dx = to_dev(torch, A), to_dev(torch, b)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    accblas.acc_trsv(ar, st, accblas.UPPER if upper else accblas.LOWER,
                     accblas.UNIT if unit else accblas.NON_UNIT, n, dA, lda, dx, 1, helper)
    torch.cuda.synchronize()
    assert int(helper[0].item()) & 0xffffffff == 0xffffffff
    got = dx.cpu().numpy()
    err = oracle.rel_error(want.astype(np.float64), got.astype(np.float64))
    # a triangular solve amplifies rounding by the factor's condition number;
    # the LU-derived factors used here keep that within ~1e2: 100 x D4 is the
    # blanket bound, 4 x the measured maximum of this (family, Ar, St, uplo)
    # the one that is asserted (tests/trsv_margins.py)
    blanket = 100 * out_tolerance(ar, st)
    trsv_margins.check(trsv_margins.key(trsv_kernel, ar, st, upper, "exact_trsv"), err, blanket)
    emulated = oracle.ref_trsv(ar, upper, unit, n, A, lda, b, 1)
    err2 = oracle.rel_error(emulated.astype(np.float64), got.astype(np.float64))
    trsv_margins.check(trsv_margins.key(trsv_kernel, ar, st, upper, "ref_trsv"), err2, blanket)


@pytest.mark.parametrize("trsv_kernel", ["rows64", "rows128"], indirect=True)
@pytest.mark.parametrize("vt", [F32, F64])
@pytest.mark.parametrize("upper", [False, True])
def test_plain_trsv_equals_accessor_bitwise(gpu, trsv_kernel, vt, upper):
    torch, accblas = gpu
    n, lda = 1500, 1536
    A = to_dev(torch, oracle.convert(triangular_system(n, lda, upper, 77), vt))
    b = make_inputs(32, n, vt)
    x1, x2 = to_dev(torch, b), to_dev(torch, b)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    uplo = accblas.UPPER if upper else accblas.LOWER
    accblas.trsv(vt, uplo, accblas.UNIT, n, A, lda, x1, 1, helper)
    accblas.acc_trsv(vt, vt, uplo, accblas.UNIT, n, A, lda, x2, 1, helper)
    torch.cuda.synchronize()
    assert torch.equal(x1, x2)


@pytest.mark.parametrize("vt", [F32, F64])
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 66, 100, 127, 128])
def test_small_systems_by_one_workgroup(gpu, vt, upper, unit, n):
    """n <= 128 runs `kernel::trsv_small`: one workgroup, forward substitution in
    the wave, no workspace. Any base alignment, row stride and increment; plain
    pointers and Acc<T, T> give the same bits; the blocked kernel (the option
    switched off) agrees within the tolerance; vector elements between the
    increments are not touched."""
    torch, accblas = gpu
    lda, incx, offset = n + 3, 3, 1
    full = triangular_system(n, lda, upper, 900 + n)
    A = oracle.convert(np.concatenate([np.zeros(offset), full]), vt)
    b = make_inputs(901, n, vt)
    xb = np.full(n * incx + offset, -7.0, dtype=A.dtype)
    xb[offset::incx][:n] = b
    want = oracle.exact_trsv(vt, upper, unit, n, A[offset:], lda, xb[offset:], incx)
    uplo = accblas.UPPER if upper else accblas.LOWER
    diag = accblas.UNIT if unit else accblas.NON_UNIT
    dA = to_dev(torch, A)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    x_plain, x_acc, x_blocked = (to_dev(torch, xb) for _ in range(3))
    accblas.trsv(vt, uplo, diag, n, dA[offset:], lda, x_plain[offset:], incx, helper)
    accblas.acc_trsv(vt, vt, uplo, diag, n, dA[offset:], lda, x_acc[offset:], incx, helper)
    accblas.set_option("trsv_small", 0)
    try:
        accblas.acc_trsv(vt, vt, uplo, diag, n, dA[offset:], lda, x_blocked[offset:], incx,
                         helper)
        torch.cuda.synchronize()
    finally:
        accblas.set_option("trsv_small", 1)
    assert int(helper[0].item()) & 0xffffffff == 0xffffffff
    assert int(helper[1].item()) == (n + 63) // 64
    assert torch.equal(x_plain, x_acc)
    got = x_plain.cpu().numpy()
    tol = 100 * out_tolerance(vt, vt)
    trsv_margins.check(trsv_margins.key("one_workgroup", vt, vt, upper, "exact_trsv"),
                       oracle.rel_error(want[::incx][:n].astype(np.float64),
                                        got[offset::incx][:n].astype(np.float64)), tol)
    assert oracle.rel_error(x_blocked.cpu().numpy()[offset::incx][:n].astype(np.float64),
                            got[offset::incx][:n].astype(np.float64)) <= tol
    untouched = np.ones(got.size, dtype=bool)
    untouched[offset::incx][:n] = False
    assert np.all(got[untouched] == -7.0)


@pytest.mark.parametrize("trsv_kernel", ["rows64", "rows128", "default"], indirect=True)
def test_trsv_manufactured_solution_strided(gpu, trsv_kernel):
    """b = T x_true formed in high precision; the solve returns x_true
    (SURVEY.md §8c C-4), with incx != 1."""
    torch, accblas = gpu
    n, lda, incx = 777, 800, 2
    A = triangular_system(n, lda, False, 5)
    T = np.tril(A.reshape(n, lda)[:, :n].astype(np.longdouble))
    x_true = np.random.default_rng(6).uniform(-1, 1, n)
    b = (T @ x_true.astype(np.longdouble)).astype(np.float64)
    xb = np.zeros(n * incx)
    xb[::incx] = b
    dx = to_dev(torch, xb)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    accblas.acc_trsv(F64, F64, accblas.LOWER, accblas.NON_UNIT, n, to_dev(torch, A), lda,
                     dx, incx, helper)
    torch.cuda.synchronize()
    got = dx.cpu().numpy()
    assert np.all(got[1::incx] == 0.0)
    assert oracle.rel_error(x_true, got[::incx]) <= 1e-11


# ------------------------------------------------------------- streams and graphs
def test_launches_on_a_side_stream_and_hipgraph_replay(gpu):
    """Launchers take an explicit stream, allocate nothing per call (TRSV: after the
    first call on that stream) and synchronise nothing, so a GEMV -> dot -> TRSV
    sequence can be captured into a HIP graph and replayed."""
    torch, accblas = gpu
    n, lda = 512, 512
    rng = np.random.default_rng(40)
    A = to_dev(torch, oracle.convert(rng.uniform(-1, 1, n * lda) / 16.0, F32))
    x = to_dev(torch, oracle.convert(rng.uniform(-1, 1, n), F32))
    res0 = to_dev(torch, oracle.convert(rng.uniform(-1, 1, n), F32))
    handle = accblas.Handle()
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")

    def sequence(stream, res, dot_out, sol):
        accblas.acc_gemv(F64, F32, n, n, 1.0, A, lda, x, 1, 1.0, res, 1, stream)
        accblas.acc_dot(handle, F64, F32, F64, n, res, 1, x, 1, dot_out, stream)
        accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n, A, lda, sol, 1, helper,
                         stream)

    side = torch.cuda.Stream()
    want_res, want_sol = res0.clone(), res0.clone()
    want_dot = torch.zeros(1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        sequence(side, want_res, want_dot, want_sol)  # also sizes the TRSV workspace
    side.synchronize()

    got_res, got_sol = res0.clone(), res0.clone()
    got_dot = torch.zeros(1, dtype=torch.float64, device="cuda")
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        sequence(torch.cuda.current_stream(), got_res, got_dot, got_sol)
    got_res.copy_(res0)
    got_sol.copy_(res0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_res, want_res)
    assert torch.equal(got_dot, want_dot)
    assert torch.equal(got_sol, want_sol)
    # replay again from the same inputs: identical bits (deterministic dot, re-armed
    # tickets and granules)
    got_res.copy_(res0)
    got_sol.copy_(res0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_res, want_res) and torch.equal(got_dot, want_dot)
    assert torch.equal(got_sol, want_sol)
    handle.close()


def test_hipgraph_replay_survives_workspace_growth(gpu):
    """A captured solve bakes the library-owned workspace of its stream into the
    kernel node. A later, larger solve on the same stream makes that workspace
    grow; the old buffer must stay allocated (retired, not freed), so replaying the
    graph afterwards still gives the bits of the eager solve."""
    torch, accblas = gpu
    n_small, n_big = 512, 9000
    A_small = to_dev(torch, oracle.convert(triangular_system(n_small, n_small, False, 401), F32))
    A_big = to_dev(torch, oracle.convert(triangular_system(n_big, n_big, False, 402), F32))
    b_small = to_dev(torch, make_inputs(403, n_small, F32))
    b_big = to_dev(torch, make_inputs(404, n_big, F32))
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()  # a fresh stream: its workspace starts small
    torch.cuda.synchronize()
    want = b_small.clone()
    with torch.cuda.stream(side):
        accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n_small, A_small, n_small,
                         want, 1, helper, side)
    side.synchronize()
    got = b_small.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n_small, A_small, n_small,
                         got, 1, helper, torch.cuda.current_stream())
    # growth during capture is refused (it would need hipMalloc + a memset)
    grown = b_big.clone()
    probe = torch.cuda.CUDAGraph()
    with pytest.raises(accblas.AccblasError) as info:
        with torch.cuda.graph(probe, stream=side):
            accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n_big, A_big, n_big,
                             grown, 1, helper, torch.cuda.current_stream())
    assert info.value.status == -2 and "capture" in str(info.value)
    torch.cuda.synchronize()
    # eager, larger solve on the same stream: the workspace grows
    grown.copy_(b_big)
    with torch.cuda.stream(side):
        accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n_big, A_big, n_big, grown,
                         1, helper, side)
    side.synchronize()
    for _ in range(2):
        got.copy_(b_small)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    assert int(helper[0].item()) & 0xffffffff == 0xffffffff
    # and the grown workspace serves the small system as well
    again = b_small.clone()
    with torch.cuda.stream(side):
        accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, n_small, A_small, n_small,
                         again, 1, helper, side)
    side.synchronize()
    assert torch.equal(again, want)
    del graph, probe
    with torch.cuda.stream(side):
        accblas.trsv_release_workspace(side)


@pytest.mark.parametrize("trsv_kernel", ["rows64", "rows128"], indirect=True)
def test_caller_owned_workspace(gpu, trsv_kernel):
    """accblas_trsv_workspace_bytes / _init / accblas_acc_trsv_ws: the caller owns
    the hand-off workspace (as the reference's caller owns `helper`,
    cuda/trsv_benchmark.cu:98); same bits as the library-owned path, usable for
    smaller systems and narrower storage, and capturable without a warm-up."""
    torch, accblas = gpu
    n, small = 3000, 700
    size = accblas.trsv_workspace_bytes(F32, n)
    def expected(words):  # header + x granules (64-word aligned) + partial-sum area
        m = n + 64  # room for the virtual unknowns of an upper solve
        return 4096 + 8 * (-(-m * words // 64) * 64 + -(-m // 128) * 15 * 128 * 2)
    assert size == expected(1)
    assert accblas.trsv_workspace_bytes(F64, n) == expected(2)
    workspace = torch.empty(size, dtype=torch.uint8, device="cuda")
    workspace.fill_(0xAB)  # garbage: init must arm it
    accblas.trsv_workspace_init(workspace)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    for m, st, upper in [(n, F32, False), (small, F32, True), (n, F16, False), (n, F32, False)]:
        A = to_dev(torch, oracle.convert(triangular_system(m, m, upper, 410 + m), st))
        b = to_dev(torch, make_inputs(411, m, st))
        uplo = accblas.UPPER if upper else accblas.LOWER
        want, got = b.clone(), b.clone()
        accblas.acc_trsv(F64, st, uplo, accblas.UNIT, m, A, m, want, 1, helper)
        accblas.acc_trsv_ws(F64, st, uplo, accblas.UNIT, m, A, m, got, 1, helper, workspace)
        torch.cuda.synchronize()
        assert int(helper[0].item()) & 0xffffffff == 0xffffffff
        assert torch.equal(got, want)
    # plain entry point
    A = to_dev(torch, oracle.convert(triangular_system(small, small, False, 420), F64))
    b = to_dev(torch, make_inputs(421, small, F64))
    want, got = b.clone(), b.clone()
    accblas.trsv(F64, accblas.LOWER, accblas.NON_UNIT, small, A, small, want, 1, helper)
    accblas.trsv_ws(F64, accblas.LOWER, accblas.NON_UNIT, small, A, small, got, 1, helper,
                    workspace)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # too small for the request / missing
    with pytest.raises(accblas.AccblasError) as info:
        accblas.acc_trsv_ws(F64, F64, accblas.LOWER, accblas.UNIT, n, A, n, got, 1, helper,
                            workspace)
    assert info.value.status == -2
    with pytest.raises(accblas.AccblasError) as info:
        accblas.acc_trsv_ws(F64, F32, accblas.LOWER, accblas.UNIT, small, A, small, got, 1,
                            helper, None)
    assert info.value.status == -3
    # captured without any eager solve on that stream
    side = torch.cuda.Stream()
    A = to_dev(torch, oracle.convert(triangular_system(small, small, False, 430), F32))
    b = to_dev(torch, make_inputs(431, small, F32))
    want, got = b.clone(), b.clone()
    accblas.acc_trsv(F64, F32, accblas.LOWER, accblas.UNIT, small, A, small, want, 1, helper)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        accblas.acc_trsv_ws(F64, F32, accblas.LOWER, accblas.UNIT, small, A, small, got, 1,
                            helper, workspace, torch.cuda.current_stream())
    got.copy_(b)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("trsv_kernel", ["rows64", "rows128"], indirect=True)
def test_trsv_concurrent_streams_and_host_threads(gpu, trsv_kernel):
    """Solves on different streams use different workspaces and may overlap on
    the device; two host threads may drive them at the same time."""
    import threading
    torch, accblas = gpu
    n = 3000
    A = to_dev(torch, oracle.convert(triangular_system(n, n, True, 91), F32))
    b = make_inputs(92, n, F32)
    want = oracle.exact_trsv(F64, True, False, n, A.cpu().numpy(), n, b, 1)
    streams = [torch.cuda.Stream() for _ in range(2)]
    helpers = [torch.zeros(2, dtype=torch.int32, device="cuda") for _ in streams]
    solutions = [[to_dev(torch, b) for _ in range(6)] for _ in streams]
    torch.cuda.synchronize()
    failures = []

    def drive(which):
        try:
            for dx in solutions[which]:
                accblas.acc_trsv(F64, F32, accblas.UPPER, accblas.NON_UNIT, n, A, n, dx, 1,
                                 helpers[which], streams[which])
        except Exception as exc:  # surfaced in the main thread
            failures.append(exc)

    threads = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not failures, failures
    for which in range(2):
        assert int(helpers[which][0].item()) & 0xffffffff == 0xffffffff
        for dx in solutions[which]:
            err = oracle.rel_error(want.astype(np.float64),
                                   dx.cpu().numpy().astype(np.float64))
            assert err <= 100 * out_tolerance(F64, F32), err


def test_trsv_workspace_grows_and_helper_reports_no_timeout(gpu):
    torch, accblas = gpu
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    for n in (64, 3000, 700, 9000):  # grow, shrink, grow again on the default stream
        A = to_dev(torch, oracle.convert(triangular_system(n, n, False, n), F64))
        b = make_inputs(50 + n, n, F64)
        want = oracle.exact_trsv(F64, False, True, n, A.cpu().numpy(), n, b, 1)
        dx = to_dev(torch, b)
        accblas.acc_trsv(F64, F64, accblas.LOWER, accblas.UNIT, n, A, n, dx, 1, helper)
        torch.cuda.synchronize()
        assert int(helper[0].item()) & 0xffffffff == 0xffffffff
        assert oracle.rel_error(want, dx.cpu().numpy()) <= 1e-11


def test_trsv_workspace_rearms_itself_between_solves(gpu):
    """One launch per solve: granules of earlier (larger, smaller, other storage
    type, other kernel family) solves stay in the workspace and must never be
    taken for this solve's; back-to-back launches
    without a host synchronisation in between must each see a re-armed state."""
    torch, accblas = gpu
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")
    plan = [(5000, F32, None), (1300, F32, None), (1300, F64, None), (5000, F16, None),
            (4224, F32, ("trsv_wide", 0)), (4224, F32, None), (1300, F32, ("trsv_wide", 2)),
            (4224, F32, ("trsv_relay", 2)), (70, F32, None), (5000, F32, None)]
    cases = {}
    for n, st, _ in plan:
        if (n, st) not in cases:
            A = oracle.convert(triangular_system(n, n, False, 7 * n), st)
            b = make_inputs(60 + n, n, st)
            cases[(n, st)] = (to_dev(torch, A), b,
                              oracle.exact_trsv(F64, False, True, n, A, n, b, 1))
    outputs = []
    for n, st, option in plan:  # no synchronisation inside this loop
        dA, b, _ = cases[(n, st)]
        dx = to_dev(torch, b)
        if option:
            accblas.set_option(*option)
        try:
            accblas.acc_trsv(F64, st, accblas.LOWER, accblas.UNIT, n, dA, n, dx, 1, helper)
            accblas.acc_trsv(F64, st, accblas.LOWER, accblas.UNIT, n, dA, n, dx, 1, helper)
        finally:
            if option:
                accblas.set_option(option[0], 1)
        outputs.append(dx)
    torch.cuda.synchronize()
    assert int(helper[0].item()) & 0xffffffff == 0xffffffff
    for (n, st, option), dx in zip(plan, outputs):
        dA, b, once = cases[(n, st)]
        # x was solved twice in place: T^-1 T^-1 b
        twice = oracle.exact_trsv(F64, False, True, n, dA.cpu().numpy(), n,
                                  once.astype(b.dtype), 1)
        err = oracle.rel_error(twice.astype(np.float64), dx.cpu().numpy().astype(np.float64))
        assert err <= 100 * out_tolerance(F64, st), (n, st, option, err)


@pytest.mark.parametrize("trsv_kernel", ["rows64", "rows128"], indirect=True)
def test_trsv_repairs_a_header_left_by_a_dead_solve_or_overwritten(gpu, trsv_kernel):
    """Every launch carries a serial number; the first workgroups of a launch
    that find the workspace header in any state but "clean" - a launch that
    never finished, bytes somebody wrote over it - repair it before a row block
    proceeds (counters reset, a fresh epoch). Nothing a predecessor left behind
    may change the result: same bits as on a clean workspace."""
    torch, accblas = gpu
    n = 3000
    A = to_dev(torch, oracle.convert(triangular_system(n, n, False, 430), F32))
    b = to_dev(torch, make_inputs(431, n, F32))
    size = accblas.trsv_workspace_bytes(F32, n)
    workspace = torch.empty(size, dtype=torch.uint8, device="cuda")
    accblas.trsv_workspace_init(workspace)
    helper = torch.zeros(2, dtype=torch.int32, device="cuda")

    def solve():
        x = b.clone()
        accblas.acc_trsv_ws(F64, F32, accblas.LOWER, accblas.UNIT, n, A, n, x, 1, helper,
                            workspace)
        assert not accblas.trsv_status(helper)  # synchronises
        return x

    clean = solve()
    want = oracle.exact_trsv(F64, False, True, n, A.cpu().numpy(), n, b.cpu().numpy(), 1)
    assert oracle.rel_error(want.astype(np.float64), clean.cpu().numpy().astype(np.float64)) \
        <= 100 * out_tolerance(F64, F32)
    header = workspace[:32].view(torch.int32)  # tickets lo/hi, finished, epoch, front
    words = workspace[4096:].view(torch.int64)
    gen = torch.Generator(device="cuda").manual_seed(5)
    # (1) random bytes over the whole 4 KiB header
    workspace[:4096] = torch.randint(0, 256, (4096,), dtype=torch.uint8, device="cuda",
                                     generator=gen)
    assert torch.equal(solve(), clean)
    assert torch.equal(solve(), clean)  # and the repaired state is a clean one
    # (2) a solve that died half way: some tickets drawn under another serial,
    # some blocks counted as finished, the front somewhere, and - worst - hand-off
    # words of the CURRENT epoch with wrong values everywhere
    epoch = int(header[3].item())
    header[0] = 5
    header[1] = 0x1234
    header[2] = 3
    header[4] = 9
    words[:] = (epoch << 32) | 0x3F800000  # every x "published" as 1.0f
    assert torch.equal(solve(), clean)
    # (3) the word of a launch that is being repaired by nobody (arming bit set
    # under a serial that is not ours)
    header[0] = -2147483648  # 0x80000000
    header[1] = 0x4321
    assert torch.equal(solve(), clean)
    # (4) all ones
    workspace[:4096] = 0xFF
    assert torch.equal(solve(), clean)
    # (5) all ones over the header of a workspace whose hand-off words carry the
    # tag the repairer will pick next (epoch 0xffffffff + 1 wraps to the first
    # epoch, 2) with wrong values: an epoch word that was written over proves
    # nothing, so the repairer clears the words
    workspace[:4096] = 0xFF
    words[:] = (2 << 32) | 0x3F800000
    assert torch.equal(solve(), clean)
    # (6) the epoch moved BACK by a scribbler while the word of a dead launch is
    # in place, words of the epoch that comes next present
    epoch = int(header[3].item())
    header[0] = 7
    header[1] = 0x2468
    header[3] = epoch - 1
    words[:] = (epoch << 32) | 0x40000000
    assert torch.equal(solve(), clean)
    assert torch.equal(solve(), clean)


def test_dot_ignores_whatever_is_in_its_scratch(gpu):
    """The dot's hand-off has no counters: a slot counts only if it carries the
    launch's serial. Garbage over the handle's whole slot area between calls
    (what a scribbling neighbour or an aborted launch leaves) changes nothing."""
    torch, accblas = gpu
    import ctypes
    lib = accblas.load()
    h = accblas.Handle()
    try:
        n = 3_000_017
        x = to_dev(torch, make_inputs(71, n, F32))
        y = to_dev(torch, make_inputs(72, n, F32))
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        accblas.acc_dot(h, F64, F32, F64, n, x, 1, y, 1, res)
        torch.cuda.synchronize()
        clean = res.item()
        scratch = ctypes.c_void_p()
        assert lib.accblas_handle_device_scratch(h.raw, ctypes.byref(scratch)) == 0
        slot_bytes = 4096 * 2 * 8  # dot_max_partials x 2 words, in front of the scalar slot
        rng = np.random.default_rng(9)
        for fill in ("random", "ones", "zeros"):
            host = {"random": rng.integers(0, 256, slot_bytes, dtype=np.uint8),
                    "ones": np.full(slot_bytes, 0xFF, dtype=np.uint8),
                    "zeros": np.zeros(slot_bytes, dtype=np.uint8)}[fill]
            assert lib.accblas_memcpy_h2d(ctypes.c_void_p(scratch.value - slot_bytes),
                                          host.ctypes.data_as(ctypes.c_void_p),
                                          slot_bytes) == 0
            for _ in range(2):
                res.fill_(-1.0)
                accblas.acc_dot(h, F64, F32, F64, n, x, 1, y, 1, res)
                torch.cuda.synchronize()
                assert res.item() == clean, fill
                assert not accblas.dot_status(h)
    finally:
        h.close()


def test_dot_status_reports_a_collector_that_gave_up_once(gpu):
    """`accblas_dot_status`: the collector of a deterministic dot leaves a sticky
    word in the handle when its bounded wait expires (and NaN in *res); the call
    reports it once and clears it. Healthy dots never set it. The word sits 32
    bytes behind the handle's scalar slot; the test plants it there, because a
    healthy device cannot be made to starve a grid for seconds."""
    torch, accblas = gpu
    import ctypes
    lib = accblas.load()
    h = accblas.Handle()
    try:
        n = 100_000
        x = to_dev(torch, make_inputs(73, n, F32))
        res = torch.zeros(1, dtype=torch.float32, device="cuda")
        accblas.acc_dot(h, F32, F32, F32, n, x, 1, x, 1, res)
        assert not accblas.dot_status(h)
        scratch = ctypes.c_void_p()
        assert lib.accblas_handle_device_scratch(h.raw, ctypes.byref(scratch)) == 0
        word = np.array([0xdeadbeef], dtype=np.uint32)
        assert lib.accblas_memcpy_h2d(ctypes.c_void_p(scratch.value + 32),
                                      word.ctypes.data_as(ctypes.c_void_p), 4) == 0
        assert accblas.dot_status(h)
        assert not accblas.dot_status(h)
        accblas.acc_dot(h, F32, F32, F32, n, x, 1, x, 1, res)
        assert not accblas.dot_status(h) and np.isfinite(res.item())
    finally:
        h.close()


@pytest.mark.parametrize("src,dst", [(F64, F32), (F64, F16), (F32, F16), (F16, F64),
                                     (F32, F64), (F16, F32)])
def test_device_convert_is_the_reference_static_cast(gpu, src, dst):
    """`accblas_convert` (convert_mtx with static_cast, matrix_helper.cuh:91-103,
    done in HBM) rounds exactly like the host conversion: single rounding, RNE,
    incl. fp16 subnormals, ties and overflow to infinity."""
    torch, accblas = gpu
    rng = np.random.default_rng(60)
    values = np.concatenate([rng.uniform(-1, 1, 20000), rng.uniform(-70000, 70000, 2000),
                             rng.uniform(-1e-7, 1e-7, 2000), 2.0 ** np.arange(-30, 17),
                             [0.0, -0.0, 65504.0, 65519.99, 65520.0, 2.0 ** -25,
                              2.0 ** -25 * (1 + 2.0 ** -30), 1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11]])
    host_src = oracle.convert(values, src)
    want = oracle.convert(host_src, dst)
    d_src = to_dev(torch, host_src)
    d_dst = torch.empty(host_src.size, dtype=accblas.torch_dtype(dst), device="cuda")
    accblas.convert(dst, d_dst, src, d_src, host_src.size)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    bits = {F16: np.uint16, F32: np.uint32, F64: np.uint64}[dst]
    assert np.array_equal(got.view(bits), want.view(bits))


def test_trsv_soak_random_solves_on_two_streams(gpu):
    """A short run of tools/soak_trsv.py: hundreds of back-to-back solves of
    random sizes / uplo / storage types on two streams, helper word and residuals
    checked along the way (the self re-arming workspace must never go stale)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools",
                        "soak_trsv.py")
    spec = importlib.util.spec_from_file_location("soak_trsv", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    module.main(iterations=800, seed=11)
