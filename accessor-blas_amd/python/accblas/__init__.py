"""ctypes binding of libaccblas_hip.so (include/accblas.h) for tests and bench.py.

This mirrors the reference's host launcher surface
(`/root/reference/cuda/gemv_kernels.cuh:136-147,168-193`,
`cuda/dot_kernels.cuh:192-263`, `cuda/trsv_kernels.cuh:455-488,918-961`) one to
one: `gemv`, `acc_gemv`, `dot`, `acc_dot`, `trsv`, `acc_trsv`, taking device
buffers (torch CUDA tensors or raw integer device pointers), sizes and strides.
PyTorch only provides device memory and streams here; all arithmetic happens
in the HIP library. There is NO fallback: if the shared library is missing or a
call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PACKAGE_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))
# The in-tree build, always. (The developer probes under tools/ that compare
# builds set `accblas.LIB_PATH` themselves before `load()`; no environment
# variable redirects this binding - bench.py and the tests measure lib/.)
LIB_PATH = os.path.join(PACKAGE_ROOT, "lib", "libaccblas_hip.so")

F16, F32, F64 = 0, 1, 2
I16, I32 = 3, 4  # integer storage: the scaled_* entry points only
UPPER, LOWER = 0, 1
NON_UNIT, UNIT = 0, 1
DOT_DETERMINISTIC, DOT_ATOMIC = 0, 1

DTYPE_NAMES = {F16: "f16", F32: "f32", F64: "f64", I16: "i16", I32: "i32"}
DTYPE_BYTES = {F16: 2, F32: 4, F64: 8, I16: 2, I32: 4}


class AccblasError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, status, message):
        super().__init__(f"accblas status {status}: {message}")
        self.status = status


_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_dbl = ctypes.c_double
_int = ctypes.c_int

# every exported symbol of include/accblas.h: name -> (restype, argtypes)
SIGNATURES = {
    "accblas_last_error_string": (ctypes.c_char_p, []),
    "accblas_version": (_int, []),
    "accblas_dtype_size": (ctypes.c_size_t, [_int]),
    "accblas_set_option": (_int, [ctypes.c_char_p, _int]),
    "accblas_device_count": (_int, [ctypes.POINTER(_int)]),
    "accblas_set_device": (_int, [_int]),
    "accblas_get_device": (_int, [ctypes.POINTER(_int)]),
    "accblas_device_info": (_int, [_int, ctypes.POINTER(_int), ctypes.POINTER(_int),
                                   ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                   ctypes.c_size_t]),
    "accblas_device_synchronize": (_int, []),
    "accblas_malloc": (_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    "accblas_free": (_int, [_vp]),
    "accblas_memcpy_h2d": (_int, [_vp, _vp, ctypes.c_size_t]),
    "accblas_memcpy_d2h": (_int, [_vp, _vp, ctypes.c_size_t]),
    "accblas_memcpy_d2d": (_int, [_vp, _vp, ctypes.c_size_t]),
    "accblas_memset": (_int, [_vp, _int, ctypes.c_size_t]),
    "accblas_stream_create": (_int, [ctypes.POINTER(_vp)]),
    "accblas_stream_destroy": (_int, [_vp]),
    "accblas_stream_synchronize": (_int, [_vp]),
    "accblas_event_create": (_int, [ctypes.POINTER(_vp)]),
    "accblas_event_destroy": (_int, [_vp]),
    "accblas_event_record": (_int, [_vp, _vp]),
    "accblas_event_synchronize": (_int, [_vp]),
    "accblas_event_elapsed_ms": (_int, [ctypes.POINTER(ctypes.c_float), _vp, _vp]),
    "accblas_handle_create": (_int, [ctypes.POINTER(_vp)]),
    "accblas_handle_destroy": (_int, [_vp]),
    "accblas_handle_num_cus": (_int, [_vp, ctypes.POINTER(_int)]),
    "accblas_handle_set_dot_mode": (_int, [_vp, _int]),
    "accblas_handle_device_scratch": (_int, [_vp, ctypes.POINTER(_vp)]),
    "accblas_handle_trsv_helper": (_int, [_vp, ctypes.POINTER(_vp)]),
    "accblas_gemv": (_int, [_int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp,
                            _i64, _vp]),
    "accblas_acc_gemv": (_int, [_int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64,
                                _dbl, _vp, _i64, _vp]),
    "accblas_dot": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "accblas_acc_dot": (_int, [_vp, _int, _int, _int, _i64, _vp, _i64, _vp, _i64, _vp,
                               _vp]),
    "accblas_dot_status": (_int, [_vp, _vp, ctypes.POINTER(_int)]),
    "accblas_scaled_acc_gemv": (_int, [_int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp,
                                       _vp, _i64, _dbl, _vp, _i64, _vp]),
    "accblas_scaled_acc_dot": (_int, [_vp, _int, _int, _int, _i64, _vp, _i64, _vp, _vp,
                                      _i64, _vp, _vp]),
    "accblas_trsv": (_int, [_int, _int, _int, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "accblas_acc_trsv": (_int, [_int, _int, _int, _int, _i64, _vp, _i64, _vp, _i64,
                                _vp, _vp]),
    "accblas_trsv_status": (_int, [_vp, _vp, ctypes.POINTER(_int)]),
    "accblas_trsv_workspace_bytes": (_int, [_int, _i64, ctypes.POINTER(ctypes.c_size_t)]),
    "accblas_trsv_workspace_init": (_int, [_vp, ctypes.c_size_t, _vp]),
    "accblas_trsv_ws": (_int, [_int, _int, _int, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                               ctypes.c_size_t, _vp]),
    "accblas_acc_trsv_ws": (_int, [_int, _int, _int, _int, _i64, _vp, _i64, _vp, _i64,
                                   _vp, _vp, ctypes.c_size_t, _vp]),
    "accblas_trsv_release_workspace": (_int, [_vp]),
    "accblas_convert": (_int, [_int, _vp, _int, _vp, _i64, _vp]),
}

_lib = None


def load():
    """Loads libaccblas_hip.so and declares every prototype. Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `make -C {PACKAGE_ROOT}` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch ships its own copy of the HIP runtime. Whichever libamdhip64 is
    # mapped first serves the whole process, and device pointers of one copy
    # mean nothing to the other ("no ROCm-capable device is detected" at the
    # first launch), so make sure torch's copy is in before ours resolves.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def library_identity():
    """(path, sha256) of the shared library this binding loads: bench.py records
    it beside its numbers."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return LIB_PATH, hashlib.sha256(f.read()).hexdigest()


def _check(status):
    if status != 0:
        raise AccblasError(status, load().accblas_last_error_string().decode())


def _ptr(buf):
    """Device pointer of a torch tensor, or the integer itself."""
    if buf is None:
        return None
    if hasattr(buf, "data_ptr"):
        return buf.data_ptr()
    return int(buf)


def dtype_of(tensor):
    """accblas_dtype of a torch tensor."""
    import torch
    return {torch.float16: F16, torch.float32: F32, torch.float64: F64}[tensor.dtype]


def torch_dtype(dt):
    import torch
    return {F16: torch.float16, F32: torch.float32, F64: torch.float64, I16: torch.int16,
            I32: torch.int32}[dt]


def _stream(stream):
    if stream is None:
        return None
    if hasattr(stream, "cuda_stream"):
        return stream.cuda_stream
    return int(stream)


def version():
    return load().accblas_version()


def set_option(name, value):
    """Tuning knobs of include/accblas.h (`accblas_set_option`)."""
    _check(load().accblas_set_option(name.encode(), int(value)))


def device_info(device=0):
    cus, wave, mem = _int(), _int(), ctypes.c_size_t()
    name = ctypes.create_string_buffer(256)
    _check(load().accblas_device_info(device, ctypes.byref(cus), ctypes.byref(wave),
                                      ctypes.byref(mem), name, 256))
    return {"num_cus": cus.value, "wavefront_size": wave.value,
            "total_mem_bytes": mem.value, "name": name.value.decode()}


def device_synchronize():
    _check(load().accblas_device_synchronize())


class Handle:
    """`myBlasHandle` (`cuda/dot_kernels.cuh:29-65`): CU count + device scratch."""

    def __init__(self):
        self._h = _vp()
        _check(load().accblas_handle_create(ctypes.byref(self._h)))

    def close(self):
        if self._h:
            _check(load().accblas_handle_destroy(self._h))
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def raw(self):
        return self._h

    @property
    def num_cus(self):
        out = _int()
        _check(load().accblas_handle_num_cus(self._h, ctypes.byref(out)))
        return out.value

    def set_dot_mode(self, mode):
        _check(load().accblas_handle_set_dot_mode(self._h, mode))

    @property
    def trsv_helper(self):
        out = _vp()
        _check(load().accblas_handle_trsv_helper(self._h, ctypes.byref(out)))
        return out.value


class Event:
    """HIP event on a given stream (`cuda_event`, `cuda/utils.cuh:113-137`)."""

    def __init__(self):
        self._e = _vp()
        _check(load().accblas_event_create(ctypes.byref(self._e)))

    def record(self, stream=None):
        _check(load().accblas_event_record(self._e, _stream(stream)))

    def synchronize(self):
        _check(load().accblas_event_synchronize(self._e))

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _check(load().accblas_event_elapsed_ms(ctypes.byref(ms), self._e, stop._e))
        return ms.value

    def __del__(self):
        try:
            if self._e:
                load().accblas_event_destroy(self._e)
        except Exception:
            pass


def gemv(vt, rows, cols, alpha, A, lda, x, incx, beta, res, incres, stream=None):
    """`gemv<ValueType>` (`cuda/gemv_kernels.cuh:136-147`)."""
    _check(load().accblas_gemv(vt, rows, cols, alpha, _ptr(A), lda, _ptr(x), incx, beta,
                               _ptr(res), incres, _stream(stream)))


def acc_gemv(ar, st, rows, cols, alpha, A, lda, x, incx, beta, res, incres,
             stream=None):
    """`acc_gemv<ArType, StType>` (`cuda/gemv_kernels.cuh:168-193`)."""
    _check(load().accblas_acc_gemv(ar, st, rows, cols, alpha, _ptr(A), lda, _ptr(x),
                                   incx, beta, _ptr(res), incres, _stream(stream)))


def dot(handle, vt, n, x, incx, y, incy, res, stream=None):
    """`dot<ValueType>` (`cuda/dot_kernels.cuh:192-206`); res = device scalar."""
    _check(load().accblas_dot(handle.raw, vt, n, _ptr(x), incx, _ptr(y), incy, _ptr(res),
                              _stream(stream)))


def scaled_acc_gemv(ar, a_st, v_st, rows, cols, alpha, A, lda, scalars, x, incx, beta, res,
                    incres, stream=None):
    """GEMV whose matrix is a `range<scaled_reduced_row_major<2, ar, a_st, 0b10>>`
    (value = storage x scalars[column]); x and res reduced_row_major<2, ar, v_st>."""
    _check(load().accblas_scaled_acc_gemv(ar, a_st, v_st, rows, cols, alpha, _ptr(A), lda,
                                          _ptr(scalars), _ptr(x), incx, beta, _ptr(res),
                                          incres, _stream(stream)))


def scaled_acc_dot(handle, ar, x_st, y_st, n, x, incx, x_scalar, y, incy, res, stream=None):
    """dot of a scaled vector (an (n, 1) `scaled_reduced_row_major` range with its one
    scalar) with a reduced_row_major vector; `res` = a device scalar of type `ar`."""
    _check(load().accblas_scaled_acc_dot(handle.raw, ar, x_st, y_st, n, _ptr(x), incx,
                                         _ptr(x_scalar), _ptr(y), incy, _ptr(res),
                                         _stream(stream)))


def acc_dot(handle, ar, st, res_t, n, x, incx, y, incy, res, stream=None):
    """`acc_dot<ArType, StType, ResType>` (`cuda/dot_kernels.cuh:224-263`)."""
    _check(load().accblas_acc_dot(handle.raw, ar, st, res_t, n, _ptr(x), incx, _ptr(y),
                                  incy, _ptr(res), _stream(stream)))


def dot_status(handle, stream=None):
    """True when the collector of a deterministic dot on `handle` gave up since
    the last call (`accblas_dot_status`; synchronises the stream)."""
    out = _int()
    _check(load().accblas_dot_status(handle.raw, _stream(stream), ctypes.byref(out)))
    return bool(out.value)


def trsv(vt, uplo, diag, n, A, lda, x, incx, helper, stream=None):
    """`trsv<ValueType>` (`cuda/trsv_kernels.cuh:455-488`)."""
    _check(load().accblas_trsv(vt, uplo, diag, n, _ptr(A), lda, _ptr(x), incx,
                               _ptr(helper), _stream(stream)))


def acc_trsv(ar, st, uplo, diag, n, A, lda, x, incx, helper, stream=None):
    """`acc_trsv<ArType, StType>` (`cuda/trsv_kernels.cuh:918-961`)."""
    _check(load().accblas_acc_trsv(ar, st, uplo, diag, n, _ptr(A), lda, _ptr(x), incx,
                                   _ptr(helper), _stream(stream)))


def trsv_status(helper, stream=None):
    """True when an internal wait of the last solve on `helper` gave up
    (`accblas_trsv_status`; synchronises the stream)."""
    out = _int()
    _check(load().accblas_trsv_status(_ptr(helper), _stream(stream), ctypes.byref(out)))
    return bool(out.value)


def _nbytes(buf):
    return buf.numel() * buf.element_size() if hasattr(buf, "numel") else 0


def trsv_workspace_bytes(st, n):
    """Bytes of a caller-owned TRSV hand-off workspace (`accblas_trsv_workspace_bytes`)."""
    out = ctypes.c_size_t()
    _check(load().accblas_trsv_workspace_bytes(st, n, ctypes.byref(out)))
    return out.value


def trsv_workspace_init(workspace, stream=None):
    """Zeroes and arms a workspace (a uint8 device tensor) once (`accblas_trsv_workspace_init`)."""
    _check(load().accblas_trsv_workspace_init(_ptr(workspace), _nbytes(workspace),
                                              _stream(stream)))


def trsv_ws(vt, uplo, diag, n, A, lda, x, incx, helper, workspace, stream=None):
    """`accblas_trsv_ws`: `trsv<ValueType>` on a caller-owned workspace."""
    _check(load().accblas_trsv_ws(vt, uplo, diag, n, _ptr(A), lda, _ptr(x), incx,
                                  _ptr(helper), _ptr(workspace), _nbytes(workspace),
                                  _stream(stream)))


def acc_trsv_ws(ar, st, uplo, diag, n, A, lda, x, incx, helper, workspace, stream=None):
    """`accblas_acc_trsv_ws`: `acc_trsv<ArType, StType>` on a caller-owned workspace."""
    _check(load().accblas_acc_trsv_ws(ar, st, uplo, diag, n, _ptr(A), lda, _ptr(x), incx,
                                      _ptr(helper), _ptr(workspace), _nbytes(workspace),
                                      _stream(stream)))


def trsv_release_workspace(stream=None):
    """Frees the library-owned workspaces of (current device, stream)."""
    _check(load().accblas_trsv_release_workspace(_stream(stream)))


def convert(dst_t, dst, src_t, src, n, stream=None):
    """Device-side `static_cast` conversion (`convert_mtx`, `cuda/matrix_helper.cuh:91-103`)."""
    _check(load().accblas_convert(dst_t, _ptr(dst), src_t, _ptr(src), n, _stream(stream)))
