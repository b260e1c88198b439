"""The TRSV setup path on ROCm: `accblas_vendor_getrf`
(rocSOLVER dgetrf, replaces the cusolverDnDgetrf call of
`/root/reference/cuda/trsv_memory.cuh:131-168`) checked against an independent
LU (scipy / LAPACK on the host). The reference factorises the row-major device
buffer READ COLUMN-MAJOR with leading dimension = the row stride, in place,
pivots discarded (`trsv_memory.cuh:150-163`): so the buffer afterwards holds the
packed LU factors of the TRANSPOSED matrix."""
import ctypes
import os

import numpy as np
import pytest
import scipy.linalg

import accblas

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vendor():
    import torch
    accblas.load()
    assert torch.cuda.is_available()
    lib = ctypes.CDLL(os.path.join(accblas.PACKAGE_ROOT, "lib", "libaccblas_vendor.so"))
    lib.accblas_vendor_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.accblas_vendor_destroy.argtypes = [ctypes.c_void_p]
    lib.accblas_vendor_getrf.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
    handle = ctypes.c_void_p()
    assert lib.accblas_vendor_create(ctypes.byref(handle)) == 0
    yield lib, handle
    lib.accblas_vendor_destroy(handle)


def factor_on_device(vendor, host, n, lda):
    import torch
    lib, handle = vendor
    dev = torch.from_numpy(host.copy()).cuda()
    info = ctypes.c_int(-99)
    status = lib.accblas_vendor_getrf(handle, n, dev.data_ptr(), lda, ctypes.byref(info))
    torch.cuda.synchronize()
    return status, info.value, dev.cpu().numpy()


@pytest.mark.parametrize("n,lda", [(300, 300), (257, 320), (64, 100), (1, 1)])
def test_getrf_matches_lapack_on_the_column_major_reading(vendor, n, lda):
    rng = np.random.default_rng(900 + n)
    buffer = rng.uniform(-1.0, 1.0, n * lda)  # row-major n x lda, as gen_mtx fills it
    status, info, got = factor_on_device(vendor, buffer, n, lda)
    assert status == 0 and info == 0
    # column-major reading with leading dimension lda = the transposed matrix
    column_major_view = buffer.reshape(n, lda)[:, :n].T
    lu, piv = scipy.linalg.lu_factor(column_major_view)
    # packed factors are stored column-major in the same buffer: element (i, j)
    # of LU sits at buffer[j * lda + i], i.e. LU^T in the row-major view
    got_lu = got.reshape(n, lda)[:, :n].T
    assert np.allclose(got_lu, lu, rtol=1e-12, atol=1e-12)
    # the padding between rows (the "stride gaps" of matrix_info) is untouched
    if lda > n:
        assert np.array_equal(got.reshape(n, lda)[:, n:], buffer.reshape(n, lda)[:, n:])
    # what the benchmark relies on (trsv_memory.cuh comment: the factors make the
    # systems well conditioned): the unit factor L of the column-major LU is the
    # row-major UPPER triangle and is bounded by 1 (partial pivoting)
    upper = np.triu(got.reshape(n, lda)[:, :n], 1)
    assert np.abs(upper).max(initial=0.0) <= 1.0 + 1e-15
    # and P L U reproduces the matrix (pivots are discarded by the setup, so use
    # LAPACK's pivots, which the comparison above has shown to be the same LU)
    L = np.tril(got_lu, -1) + np.eye(n)
    U = np.triu(got_lu)
    perm = np.arange(n)
    for i, p in enumerate(piv):
        perm[[i, p]] = perm[[p, i]]
    assert np.allclose((L @ U), column_major_view[perm], rtol=1e-11, atol=1e-11)


def test_getrf_reports_a_singular_matrix(vendor):
    n = 64
    buffer = np.random.default_rng(7).uniform(-1.0, 1.0, n * n)
    buffer.reshape(n, n)[5, :] = 0.0  # a zero COLUMN of the column-major view
    status, info, _ = factor_on_device(vendor, buffer, n, n)
    assert status == 0
    assert info == 6  # 1-based index of the first zero pivot


def test_getrf_rejects_bad_arguments(vendor):
    lib, handle = vendor
    info = ctypes.c_int(0)
    assert lib.accblas_vendor_getrf(handle, 10, None, 10, ctypes.byref(info)) == -3
    assert lib.accblas_vendor_getrf(handle, 10, ctypes.c_void_p(0x1000), 5,
                                    ctypes.byref(info)) == -2
    assert lib.accblas_vendor_getrf(handle, 0, None, 0, ctypes.byref(info)) == 0


def test_trsv_benchmark_setup_rejects_a_singular_matrix(tmp_path):
    """The holder's generating constructor checks `info` (the reference ignores
    it, trsv_memory.cuh:150-166): a driver fed a singular matrix must fail
    loudly. Exercised through a tiny program built on hip/trsv_memory.hpp."""
    import subprocess
    pkg = accblas.PACKAGE_ROOT
    src = tmp_path / "singular.cpp"
    src.write_text(r'''
#include <cstdio>
#include <stdexcept>
#include "trsv_memory.hpp"
int main() {
    auto gen_mtx = [](matrix_info info) {
        Memory<double> m(Memory<double>::Device::cpu, info.get_1d_size());
        for (long i = 0; i < (long)info.get_1d_size(); ++i) m.data()[i] = 0.25 + (i % 7);
        for (long j = 0; j < info.size[1]; ++j) m.data()[3 * info.stride + j] = 0.0;
        return m;
    };
    auto gen_x = [](matrix_info info) {
        Memory<double> m(Memory<double>::Device::cpu, info.get_1d_size());
        for (long i = 0; i < (long)info.get_1d_size(); ++i) m.data()[i] = 1.0;
        return m;
    };
    try {
        TrsvMemory<double> data(32, gen_mtx, gen_x);
    } catch (const std::runtime_error &e) {
        std::printf("caught: %s\n", e.what());
        return 0;
    }
    std::printf("no exception\n");
    return 1;
}
''')
    exe = tmp_path / "singular"
    root = os.path.dirname(pkg)
    subprocess.run(["g++", "-std=c++14", "-O1", str(src), "-o", str(exe),
                    "-I", os.path.join(pkg, "include"), "-I", os.path.join(root, "include"),
                    "-I", os.path.join(pkg, "hip"), "-L", os.path.join(pkg, "lib"),
                    "-laccblas_vendor", "-laccblas_hip",
                    "-Wl,-rpath," + os.path.join(pkg, "lib"),
                    "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "LU factorisation failed, info = " in out.stdout
