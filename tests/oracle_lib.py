"""ctypes wrapper of oracle/liboracle.so (TEST INFRASTRUCTURE).

The oracle is the CPU restatement of the reference's path (see the header of
oracle/accblas_oracle.cpp). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

F16, F32, F64 = 0, 1, 2
NP_DTYPE = {F16: np.float16, F32: np.float32, F64: np.float64}
CODE_OF = {np.dtype(np.float16): F16, np.dtype(np.float32): F32,
           np.dtype(np.float64): F64}

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_dbl = ctypes.c_double
_int = ctypes.c_int
_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = ctypes.CDLL(LIB_PATH)
    sig = {
        "oracle_engine_create": (_vp, [ctypes.c_uint32]),
        "oracle_engine_destroy": (None, [_vp]),
        "oracle_gen_mtx": (_int, [_vp, _i64, _i64, _i64, _vp]),
        "oracle_engine_discard": (None, [_vp, _i64]),
        "oracle_write_random": (None, [_vp, _i64, _i64, _i64, _vp]),
        "oracle_convert": (_int, [_int, _vp, _int, _vp, _i64]),
        "oracle_reduce": (_dbl, [_i64, _i64, _vp, _int]),
        "oracle_compare": (_dbl, [_i64, _i64, _vp, _int, _vp, _vp]),
        "oracle_ref_gemv": (_int, [_int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64,
                                   _dbl, _vp, _i64, _int]),
        "oracle_exact_gemv": (_int, [_int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64,
                                     _dbl, _vp, _i64]),
        "oracle_ref_dot": (_int, [_int, _int, _int, _i64, _vp, _i64, _vp, _i64, _vp,
                                  _i64, _int]),
        "oracle_exact_dot": (_int, [_int, _int, _i64, _vp, _i64, _vp, _i64,
                                    ctypes.POINTER(_dbl), ctypes.POINTER(_dbl)]),
        "oracle_ref_trsv": (_int, [_int, _int, _int, _int, _i64, _vp, _i64, _vp, _i64,
                                   _int]),
        "oracle_exact_trsv": (_int, [_int, _int, _int, _int, _i64, _vp, _i64, _vp,
                                     _i64]),
        "oracle_accessor_gemv_rows": (_int, [_int, _int, _i64, _i64, _i64, _dbl, _vp,
                                             _i64, _vp, _i64, _dbl, _vp, _i64]),
        "oracle_accessor_dot_chunk": (_int, [_int, _int, _i64, _i64, _vp, _i64, _vp,
                                             _i64, ctypes.POINTER(_dbl)]),
    }
    for name, (restype, argtypes) in sig.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _p(arr):
    return arr.ctypes.data_as(_vp)


def code(arr):
    return CODE_OF[arr.dtype]


class Engine:
    """std::default_random_engine(seed) + uniform_real_distribution<double>(-1, 1)
    (`/root/reference/cuda/gemv_benchmark.cu:81-83`)."""

    def __init__(self, seed=42):
        self._e = load().oracle_engine_create(seed)

    def gen_mtx(self, rows, cols, stride=None):
        """`gen_mtx` (`cuda/matrix_helper.cuh:28-51`); returns the flat buffer of
        rows*stride doubles (stride gaps are zero here, untouched there)."""
        stride = cols if stride is None else stride
        out = np.zeros(rows * stride, dtype=np.float64)
        status = load().oracle_gen_mtx(self._e, rows, cols, stride, _p(out))
        if status != 0:
            raise RuntimeError("Wrong use of stride")
        return out

    def discard(self, count):
        """Draws `count` values as gen_mtx would, without storing them."""
        load().oracle_engine_discard(self._e, count)

    def write_random(self, out, rows, cols, stride=None):
        """`write_random` (`cuda/matrix_helper.cuh:66-76`)."""
        stride = cols if stride is None else stride
        load().oracle_write_random(self._e, rows, cols, stride, _p(out))

    def __del__(self):
        try:
            load().oracle_engine_destroy(self._e)
        except Exception:
            pass


def convert(src, dst_code):
    """`convert_mtx` with static_cast (`cuda/matrix_helper.cuh:91-103`)."""
    src = np.ascontiguousarray(src)
    dst = np.empty(src.shape, dtype=NP_DTYPE[dst_code])
    load().oracle_convert(dst_code, _p(dst), code(src), _p(src), src.size)
    return dst


def reduce(values, mode=0, stride=1):
    """`reduce` (`cuda/utils.cuh:281-300`); mode 0: a+b, 1: |a|+|b|."""
    tmp = np.array(values, dtype=np.float64, copy=True)
    n = (tmp.size + stride - 1) // stride
    return load().oracle_reduce(n, stride, _p(tmp), mode)


def compare(ref, other, stride=1):
    """`compare` (`cuda/utils.cuh:315-332`): sum |ref - other|."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    other = np.ascontiguousarray(other)
    n = (ref.size + stride - 1) // stride
    tmp = np.empty(ref.size, dtype=np.float64)
    return load().oracle_compare(n, stride, _p(ref), code(other), _p(other), _p(tmp))


def rel_error(ref, other):
    """The reference's `--error` metric: sum|ref-res| / sum|ref|
    (`cuda/gemv_benchmark.cu:101-124,219-232`)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    return compare(ref, other) / reduce(ref, mode=1)


def ref_gemv(ar, rows, cols, alpha, A, lda, x, incx, beta, res, incres, use_fma=True):
    """Order-faithful emulation of `kernel::acc_gemv<512>`; returns the new res."""
    out = np.array(res, copy=True)
    status = load().oracle_ref_gemv(ar, code(A), rows, cols, alpha, _p(A), lda, _p(x),
                                    incx, beta, _p(out), incres, int(use_fma))
    assert status == 0
    return out


def exact_gemv(ar, rows, cols, alpha, A, lda, x, incx, beta, res, incres):
    out = np.array(res, copy=True)
    status = load().oracle_exact_gemv(ar, code(A), rows, cols, alpha, _p(A), lda, _p(x),
                                      incx, beta, _p(out), incres)
    assert status == 0
    return out


def ref_dot(ar, res_code, n, x, incx, y, incy, num_blocks, use_fma=True):
    out = np.zeros(1, dtype=NP_DTYPE[res_code])
    status = load().oracle_ref_dot(ar, code(x), res_code, n, _p(x), incx, _p(y), incy,
                                   _p(out), num_blocks, int(use_fma))
    assert status == 0
    return out[0]


def exact_dot(ar, n, x, incx, y, incy):
    """Returns (sum, Q): long-double sum rounded to double and
    Q = sqrt(sum (x_i y_i)^2), the natural scale of the sum."""
    total, scale = _dbl(), _dbl()
    status = load().oracle_exact_dot(ar, code(x), n, _p(x), incx, _p(y), incy,
                                     ctypes.byref(total), ctypes.byref(scale))
    assert status == 0
    return total.value, scale.value


def ref_trsv(ar, upper, unit, n, A, lda, x, incx, use_fma=True):
    out = np.array(x, copy=True)
    status = load().oracle_ref_trsv(ar, code(A), int(upper), int(unit), n, _p(A), lda,
                                    _p(out), incx, int(use_fma))
    assert status == 0
    return out


def exact_trsv(ar, upper, unit, n, A, lda, x, incx):
    out = np.array(x, copy=True)
    status = load().oracle_exact_trsv(ar, code(A), int(upper), int(unit), n, _p(A), lda,
                                      _p(out), incx)
    assert status == 0
    return out


def accessor_gemv_rows(ar, row_begin, row_end, cols, alpha, A, lda, x, incx, beta, res,
                       incres):
    """In-place CPU GEMV of rows [row_begin, row_end) through the host accessor."""
    status = load().oracle_accessor_gemv_rows(ar, code(A), row_begin, row_end, cols,
                                              alpha, _p(A), lda, _p(x), incx, beta,
                                              _p(res), incres)
    assert status == 0


def accessor_dot_chunk(ar, begin, end, x, incx, y, incy):
    out = _dbl()
    status = load().oracle_accessor_dot_chunk(ar, code(x), begin, end, _p(x), incx,
                                              _p(y), incy, ctypes.byref(out))
    assert status == 0
    return out.value
