"""ctypes binding of libaccblas_rccl.so (include/accblas_comm.h): the C-level
multi-GPU context (RCCL communicator + sharded dot / row-block GEMV).

`bench.py` and `multi_gpu.py` drive the ranks through torch.distributed; this
module is the same composition for hosts without PyTorch (it is what
`bin/multi_gpu_benchmark` links). No fallback: a missing library raises.
"""
import ctypes
import os

from . import (AccblasError, PACKAGE_ROOT, _dbl, _i64, _int, _ptr, _stream, _vp, load)

LIB_PATH = os.path.join(PACKAGE_ROOT, "lib", "libaccblas_rccl.so")
ID_BYTES = 128
ERR_COMM = -5

_pp = ctypes.POINTER(_vp)
_pi = ctypes.POINTER(_int)
_pi64 = ctypes.POINTER(_i64)

# every exported symbol of include/accblas_comm.h: name -> (restype, argtypes)
SIGNATURES = {
    "accblas_comm_last_error_string": (ctypes.c_char_p, []),
    "accblas_comm_unique_id": (_int, [_vp]),
    "accblas_comm_create": (_int, [_pp, _int, _int, _vp]),
    "accblas_comm_create_all": (_int, [_pp, _int, _pi]),
    "accblas_comm_destroy": (_int, [_vp]),
    "accblas_comm_rank": (_int, [_vp, _pi]),
    "accblas_comm_world_size": (_int, [_vp, _pi]),
    "accblas_comm_device": (_int, [_vp, _pi]),
    "accblas_comm_group_begin": (_int, []),
    "accblas_comm_group_end": (_int, []),
    "accblas_comm_shard": (_int, [_vp, _i64, _i64, _pi64, _pi64]),
    "accblas_shard_bounds": (_int, [_i64, _int, _int, _i64, _pi64, _pi64]),
    "accblas_comm_acc_dot": (_int, [_vp, _vp, _int, _int, _int, _i64, _vp, _i64, _vp,
                                    _i64, _vp, _vp]),
    "accblas_comm_acc_gemv": (_int, [_vp, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp,
                                     _i64, _dbl, _vp, _i64, _vp, _vp]),
}

_lib = None


def load_comm():
    """Loads libaccblas_rccl.so (after libaccblas_hip.so, and so after torch's
    copies of the HIP runtime and RCCL when torch is installed)."""
    global _lib
    if _lib is not None:
        return _lib
    load()
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `make -C {PACKAGE_ROOT}`; "
            "there is no fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(status):
    if status != 0:
        raise AccblasError(status, load_comm().accblas_comm_last_error_string().decode())


def unique_id():
    """Rank 0: the 128 bytes every rank passes to `Comm.create`."""
    buf = ctypes.create_string_buffer(ID_BYTES)
    _check(load_comm().accblas_comm_unique_id(ctypes.cast(buf, _vp)))
    return buf.raw


class Comm:
    """One rank of an RCCL communicator bound to a device."""

    def __init__(self, raw):
        self._c = raw

    @classmethod
    def create(cls, world_size, rank, uid):
        """Collective; binds the calling thread's current device."""
        raw = _vp()
        buf = ctypes.create_string_buffer(bytes(uid), ID_BYTES)
        _check(load_comm().accblas_comm_create(ctypes.byref(raw), world_size, rank,
                                               ctypes.cast(buf, _vp)))
        return cls(raw)

    @classmethod
    def create_all(cls, n_gpus, devices=None):
        """One process driving n_gpus devices: returns the n_gpus ranks."""
        raws = (_vp * n_gpus)()
        devs = (_int * n_gpus)(*devices) if devices is not None else None
        _check(load_comm().accblas_comm_create_all(raws, n_gpus, devs))
        return [cls(_vp(r)) for r in raws]

    def close(self):
        if self._c:
            _check(load_comm().accblas_comm_destroy(self._c))
            self._c = _vp()

    def _int_query(self, fn):
        out = _int()
        _check(fn(self._c, ctypes.byref(out)))
        return out.value

    @property
    def rank(self):
        return self._int_query(load_comm().accblas_comm_rank)

    @property
    def world_size(self):
        return self._int_query(load_comm().accblas_comm_world_size)

    @property
    def device(self):
        return self._int_query(load_comm().accblas_comm_device)

    def shard(self, n, align=1):
        lo, hi = _i64(), _i64()
        _check(load_comm().accblas_comm_shard(self._c, n, align, ctypes.byref(lo),
                                              ctypes.byref(hi)))
        return lo.value, hi.value

    def acc_dot(self, handle, ar, st, res_t, n_local, x, incx, y, incy, res,
                stream=None):
        """Chunk dot + one all-reduce of the arithmetic-type scalar."""
        _check(load_comm().accblas_comm_acc_dot(
            self._c, handle.raw, ar, st, res_t, n_local, _ptr(x), incx, _ptr(y), incy,
            _ptr(res), _stream(stream)))

    def acc_gemv(self, ar, st, rows_local, cols, alpha, A, lda, x, incx, beta,
                 res_local, incres, res_full=None, stream=None):
        """Row-block GEMV; `res_full` adds the all-gather of the slices."""
        _check(load_comm().accblas_comm_acc_gemv(
            self._c, ar, st, rows_local, cols, alpha, _ptr(A), lda, _ptr(x), incx, beta,
            _ptr(res_local), incres, _ptr(res_full), _stream(stream)))


def shard_bounds(n, world_size, rank, align=1):
    """`accblas_shard_bounds`: rank's [lo, hi) of a balanced contiguous partition."""
    lo, hi = _i64(), _i64()
    _check(load_comm().accblas_shard_bounds(n, world_size, rank, align,
                                            ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def group_begin():
    _check(load_comm().accblas_comm_group_begin())


def group_end():
    _check(load_comm().accblas_comm_group_end())
