"""CPU tests of the oracle (runs without a GPU): the restatement is pinned against
the golden fixtures under tests/golden/ (known answers derived from libstdc++'s
generator semantics in SURVEY.md §8c C-4, numpy's fp16 rounding) and against
structural known answers. The reference itself ships no tests or fixtures."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as oracle
from oracle_lib import F16, F32, F64

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gen_golden():
    with open(os.path.join(GOLDEN, "generator.json")) as f:
        return json.load(f)


def test_generator_first_values(gen_golden):
    values = oracle.Engine(42).gen_mtx(1, 6)
    assert values.tolist() == gen_golden["first_values"]
    as_float = oracle.convert(values[:3], F32)
    assert as_float.tolist() == [np.float32(v) for v in gen_golden["first_values_as_float"]]


def test_generator_gemv_draw_order(gen_golden):
    """A (max^2) -> x -> res (`cuda/gemv_memory.cuh:45-47`): x[0] is draw number
    max_size^2, res[0] follows the max_size values of x."""
    n = gen_golden["gemv_max_size"]
    engine = oracle.Engine(42)
    head = engine.gen_mtx(1, 3)
    assert head.tolist() == gen_golden["first_values"][:3]
    engine.discard(n * n - 3)
    x = engine.gen_mtx(n, 1)
    res = engine.gen_mtx(1, 1)
    assert x[0] == gen_golden["gemv_x0"]
    assert res[0] == gen_golden["gemv_res_init0"]


def test_generator_stride_and_error():
    engine = oracle.Engine(42)
    flat = engine.gen_mtx(2, 3, stride=5)
    dense = oracle.Engine(42).gen_mtx(2, 3)
    assert flat.reshape(2, 5)[:, :3].reshape(-1).tolist() == dense.tolist()
    assert np.all(flat.reshape(2, 5)[:, 3:] == 0.0)  # gaps untouched
    with pytest.raises(RuntimeError, match="Wrong use of stride"):
        engine.gen_mtx(2, 3, stride=2)


def test_config_c1_dot_fp64_on_cpu(gen_golden):
    """BASELINE config 1: dot n=1M fp64/fp64 with the reference generator."""
    n = gen_golden["dot_n"]
    engine = oracle.Engine(42)
    x = engine.gen_mtx(n, 1)
    y = engine.gen_mtx(n, 1)
    assert x[-1] == pytest.approx(gen_golden["dot_x_last"], rel=1e-13)
    assert y[0] == gen_golden["dot_y_first"]
    assert y[-1] == gen_golden["dot_y_last"]
    total, scale = oracle.exact_dot(F64, n, x, 1, y, 1)
    assert total == pytest.approx(gen_golden["dot_fp64"], rel=1e-13)
    assert scale == pytest.approx(np.sqrt(n) / 3, rel=0.01)
    # order-faithful emulation of kernel::dot<1024> on an A100-sized grid
    emulated = oracle.ref_dot(F64, F64, n, x, 1, y, 1, 108 * 32)
    assert emulated == pytest.approx(gen_golden["dot_fp64"], rel=1e-13)
    xs, ys = oracle.convert(x, F32), oracle.convert(y, F32)
    total32, _ = oracle.exact_dot(F64, n, xs, 1, ys, 1)
    assert total32 == pytest.approx(gen_golden["dot_of_fp32_rounded_copies"], rel=1e-13)


def test_fp16_conversion_bit_patterns():
    with open(os.path.join(GOLDEN, "fp16_patterns.json")) as f:
        fixture = json.load(f)
    values = np.array([struct.unpack(">d", bytes.fromhex(h))[0]
                       for h in fixture["f64_inputs_hex"]])
    got = oracle.convert(values, F16).view(np.uint16)
    assert got.tolist() == fixture["f64_to_f16_bits"]
    singles = np.array(fixture["f32_inputs_bits"], dtype=np.uint32).view(np.float32)
    got32 = oracle.convert(singles, F16).view(np.uint16)
    assert got32.tolist() == fixture["f32_to_f16_bits"]
    # widening is exact
    halves = np.array(fixture["f64_to_f16_bits"], dtype=np.uint16).view(np.float16)
    finite = np.isfinite(halves)
    assert np.array_equal(oracle.convert(halves, F64)[finite],
                          halves.astype(np.float64)[finite])


def test_reference_reduce_pairing():
    """`reduce` (`cuda/utils.cuh:281-300`) pairs element r with r + ceil(len/2)."""
    assert oracle.reduce([5.0]) == 5.0
    assert oracle.reduce([1.0, 2.0]) == 3.0
    assert oracle.reduce([1.0, 2.0, 3.0]) == 6.0
    # pairing (0,2),(1,3): (1e16 - 1e16) + (1 + 1) = 2; a serial sum gives 1
    assert oracle.reduce([1e16, 1.0, -1e16, 1.0]) == 2.0
    for n in (5, 8, 17, 64, 1000, 1025):
        values = np.arange(1, n + 1, dtype=np.float64)
        assert oracle.reduce(values) == n * (n + 1) / 2
        assert oracle.reduce(-values, mode=1) == n * (n + 1) / 2
    # norm mode on a single element: op(tmp[0], {}) = |a| + |0|
    assert oracle.reduce([-3.0], mode=1) == 3.0


def test_reference_compare():
    ref = np.array([1.0, -2.0, 3.5, 0.0, 8.0])
    other = np.array([1.5, -2.0, 3.0, 1.0, 6.0], dtype=np.float32)
    assert oracle.compare(ref, other) == 0.5 + 0.0 + 0.5 + 1.0 + 2.0
    assert oracle.rel_error(ref, other) == 4.0 / 14.5


@pytest.mark.parametrize("ar,st,tol", [(F64, F64, 1e-14), (F64, F32, 1e-7), (F32, F32, 5e-6),
                                       (F64, F16, 1e-3), (F32, F16, 1e-3)])
def test_emulated_gemv_agrees_with_exact(ar, st, tol):
    rng = np.random.default_rng(3)
    rows, cols, lda = 50, 1300, 1400
    A = oracle.convert(rng.uniform(-1, 1, rows * lda), st)
    x = oracle.convert(rng.uniform(-1, 1, cols), st)
    res = oracle.convert(rng.uniform(-1, 1, rows), st)
    exact = oracle.exact_gemv(ar, rows, cols, 1.0, A, lda, x, 1, 1.0, res, 1)
    for fma in (True, False):
        emulated = oracle.ref_gemv(ar, rows, cols, 1.0, A, lda, x, 1, 1.0, res, 1,
                                   use_fma=fma)
        assert oracle.rel_error(exact.astype(np.float64),
                                emulated.astype(np.float64)) <= tol


def test_published_error_magnitudes_gemv():
    """The reference's plots: Acc<fp64,fp32> GEMV error vs the fp64
    kernel on unrounded data is ~3.8-4.2e-8 flat in n; pure fp32 ~8.4e-8..1.5e-7.
    The emulation of the CUDA kernels on reference-generated data reproduces those
    magnitudes (not the individual plot points, which used other input data)."""
    n = 1300
    engine = oracle.Engine(42)
    A = engine.gen_mtx(n, n)
    x = engine.gen_mtx(n, 1)
    res = engine.gen_mtx(n, 1)
    truth = oracle.ref_gemv(F64, n, n, 1.0, A, n, x, 1, 1.0, res, 1)  # "GEMV fp64"
    A32, x32, r32 = (oracle.convert(v, F32) for v in (A, x, res))
    acc = oracle.ref_gemv(F64, n, n, 1.0, A32, n, x32, 1, 1.0, r32, 1)
    f32 = oracle.ref_gemv(F32, n, n, 1.0, A32, n, x32, 1, 1.0, r32, 1)
    err_acc = oracle.rel_error(truth, acc.astype(np.float64))
    err_f32 = oracle.rel_error(truth, f32.astype(np.float64))
    assert 2.5e-8 <= err_acc <= 5.5e-8, err_acc
    assert 6e-8 <= err_f32 <= 2e-7, err_f32
    assert err_f32 > 1.5 * err_acc


def _triangular(n, lda, upper, unit, seed):
    rng = np.random.default_rng(seed)
    full = rng.uniform(-1, 1, (n, lda))
    scale = 1.0 / max(n, 1)  # keeps the factor well conditioned
    tri = (np.triu(full[:, :n], 1) if upper else np.tril(full[:, :n], -1)) * scale * 8
    junk = np.tril(full[:, :n], -1) if upper else np.triu(full[:, :n], 1)
    diag = rng.uniform(1.0, 2.0, n)
    full[:, :n] = tri + junk + np.diag(diag)
    return full, (tri + (np.eye(n) if unit else np.diag(diag)))


@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("n", [1, 31, 32, 45, 70, 100, 129])
def test_trsv_manufactured_solutions(upper, unit, n):
    """b = T x_true in extended precision; both the emulation of the CUDA kernels
    (32-wide tiles, ragged last/top block) and the semantic oracle recover x_true."""
    lda = n + 7
    full, T = _triangular(n, lda, upper, unit, 100 + n)
    x_true = np.random.default_rng(n).uniform(-1, 1, n)
    b = (T.astype(np.longdouble) @ x_true.astype(np.longdouble)).astype(np.float64)
    flat = full.reshape(-1)
    got_ref = oracle.ref_trsv(F64, upper, unit, n, flat, lda, b, 1)
    got_exact = oracle.exact_trsv(F64, upper, unit, n, flat, lda, b, 1)
    assert oracle.rel_error(x_true, got_ref) <= 1e-13
    assert oracle.rel_error(x_true, got_exact) <= 1e-13
    # strided x
    xb = np.zeros(3 * n)
    xb[::3] = b
    got_strided = oracle.ref_trsv(F64, upper, unit, n, flat, lda, xb, 3)
    assert np.array_equal(got_strided[::3], got_ref)
    assert np.all(got_strided[1::3] == 0) and np.all(got_strided[2::3] == 0)


@pytest.mark.parametrize("ar,st,tol", [(F64, F32, 1e-6), (F32, F32, 5e-5), (F64, F16, 2e-2)])
def test_trsv_reduced_storage(ar, st, tol):
    n, lda = 200, 200
    full, _ = _triangular(n, lda, True, True, 9)
    A = oracle.convert(full.reshape(-1), st)
    b = oracle.convert(np.random.default_rng(1).uniform(-1, 1, n), st)
    ref = oracle.ref_trsv(ar, True, True, n, A, lda, b, 1)
    exact = oracle.exact_trsv(ar, True, True, n, A, lda, b, 1)
    assert oracle.rel_error(exact.astype(np.float64), ref.astype(np.float64)) <= tol


def test_dot_emulation_grid_and_types():
    rng = np.random.default_rng(8)
    n = 300_000
    x = oracle.convert(rng.uniform(-1, 1, n), F16)
    y = oracle.convert(rng.uniform(-1, 1, n), F16)
    exact, scale = oracle.exact_dot(F32, n, x, 1, y, 1)
    for blocks in (1, 7, 256):
        got = oracle.ref_dot(F32, F32, n, x, 1, y, 1, blocks)
        assert abs(float(got) - exact) <= 64 * 2.0 ** -24 * scale
    # ResType != ArType (cast_result): fp64 arithmetic, fp32 result
    got = oracle.ref_dot(F64, F32, n, x, 1, y, 1, 64)
    assert got == np.float32(oracle.ref_dot(F64, F64, n, x, 1, y, 1, 64))
    assert oracle.ref_dot(F64, F64, 0, x, 1, y, 1, 4) == 0.0
