"""C-ABI checks that need no GPU: the library loads, exports exactly what
include/accblas.h declares, the Python prototypes cover the same set, and
argument validation (which happens before any HIP call) returns the documented
status codes."""
import ctypes
import os
import re
import subprocess

import pytest

import accblas

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HEADER = os.path.join(REPO, "include", "accblas.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(accblas_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = accblas.load()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in accblas.h but not exported"
    exported = subprocess.run(["nm", "-D", "--defined-only", accblas.LIB_PATH],
                              check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (accblas_[a-z0-9_]+)\b", exported))
    assert exported == set(names), exported.symmetric_difference(names)


def test_python_prototypes_match_header():
    assert sorted(accblas.SIGNATURES) == declared_symbols()


def test_header_is_plain_c():
    """The boundary must be consumable from C (and so from cgo/JNI/ctypes)."""
    src = os.path.join(REPO, "gpurun_out", "abi_check.c")
    os.makedirs(os.path.dirname(src), exist_ok=True)
    with open(src, "w") as f:
        f.write('#include "accblas.h"\nint main(void){return accblas_version() == 0;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-c", src,
                    "-I", os.path.join(REPO, "include"), "-o", src + ".o"], check=True)


def test_host_only_entry_points():
    lib = accblas.load()
    assert lib.accblas_version() == 100
    assert [lib.accblas_dtype_size(t) for t in (0, 1, 2)] == [2, 4, 8]
    assert isinstance(lib.accblas_last_error_string(), bytes)


def test_argument_validation_statuses():
    lib = accblas.load()
    dummy = ctypes.c_void_p(0x1000)
    # negative size
    assert lib.accblas_acc_gemv(2, 1, -1, 4, 1.0, dummy, 4, dummy, 1, 1.0, dummy, 1,
                                None) == -2
    assert b"negative" in lib.accblas_last_error_string()
    # row stride smaller than cols ("Wrong use of stride", matrix_helper.cuh:31-33)
    assert lib.accblas_gemv(2, 4, 8, 1.0, dummy, 4, dummy, 1, 1.0, dummy, 1, None) == -2
    # NULL operands
    assert lib.accblas_gemv(2, 4, 4, 1.0, None, 4, dummy, 1, 1.0, dummy, 1, None) == -3
    # dot without a handle
    assert lib.accblas_dot(None, 2, 10, dummy, 1, dummy, 1, dummy, None) == -3
    # trsv: helper is required, lda >= n, increments >= 1
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 8, dummy, 1, None, None) == -3
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 4, dummy, 1, dummy, None) == -2
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 8, dummy, 0, dummy, None) == -2
    assert lib.accblas_trsv(2, 7, 1, 8, dummy, 8, dummy, 1, dummy, None) == -2
    with pytest.raises(accblas.AccblasError) as info:
        accblas.gemv(accblas.F64, 4, 4, 1.0, 0x1000, 2, 0x1000, 1, 1.0, 0x1000, 1)
    assert info.value.status == -2 and "stride" in str(info.value)


def test_options_are_validated():
    lib = accblas.load()
    assert lib.accblas_set_option(b"no_such_option", 1) == -2
    assert lib.accblas_set_option(None, 1) == -2
    assert lib.accblas_set_option(b"gemv_skew", 3) == -2
    assert lib.accblas_set_option(b"trsv_split", 0) == -2  # removed in round 2
    assert lib.accblas_set_option(b"trsv_wide_split_from", -1) == -2
    assert lib.accblas_set_option(b"trsv_wide_split_from", 0) == 0
    for name, default in ((b"gemv_skew", 1), (b"trsv_relay", 1), (b"trsv_wide", 1),
                          (b"trsv_prefetch", 1), (b"trsv_small", 1), (b"trsv_vector_rows", 1),
                          (b"gemv_vector_rows", 1), (b"trsv_wide_persistent", 0)):
        assert lib.accblas_set_option(name, 3) == -2
        assert lib.accblas_set_option(name, default) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(accblas, "_lib", None)
    monkeypatch.setattr(accblas, "LIB_PATH", "/nonexistent/libaccblas_hip.so")
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        accblas.load()


# ---- the other two boundary headers ------------------------------------------

def header_symbols(name):
    text = open(os.path.join(REPO, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(accblas_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(library):
    out = subprocess.run(["nm", "-D", "--defined-only", library], check=True,
                         capture_output=True, text=True).stdout
    return set(re.findall(r"\bT (accblas_[a-z0-9_]+)\b", out))


@pytest.mark.parametrize("header, library", [
    ("accblas_vendor.h", "libaccblas_vendor.so"),
    ("accblas_comm.h", "libaccblas_rccl.so"),
])
def test_side_libraries_export_their_headers(header, library):
    names = header_symbols(header)
    assert names, header
    path = os.path.join(accblas.PACKAGE_ROOT, "lib", library)
    assert exported_symbols(path) == set(names)


def test_comm_prototypes_and_host_side_validation():
    from accblas import comm, multi_gpu
    lib = comm.load_comm()
    assert sorted(comm.SIGNATURES) == header_symbols("accblas_comm.h")
    # the partition is the one the torch.distributed layer uses
    for n, world in [(0, 1), (10, 3), (7, 8), (1 << 28, 8), (32768, 6)]:
        pieces = [comm.shard_bounds(n, world, r) for r in range(world)]
        assert pieces == [multi_gpu.shard_bounds(n, world, r) for r in range(world)]
    # aligned boundaries still tile [0, n) and interior edges are multiples of 8
    pieces = [comm.shard_bounds(1000003, 8, r, 8) for r in range(8)]
    assert pieces[0][0] == 0 and pieces[-1][1] == 1000003
    assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    assert all(p[1] % 8 == 0 for p in pieces[:-1])
    # validation happens before RCCL / HIP are touched
    dummy = ctypes.c_void_p(0x1000)
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    assert lib.accblas_shard_bounds(10, 0, 0, 1, ctypes.byref(lo), ctypes.byref(hi)) == -2
    assert lib.accblas_shard_bounds(10, 2, 2, 1, ctypes.byref(lo), ctypes.byref(hi)) == -2
    assert lib.accblas_comm_create(None, 1, 0, dummy) == -3
    raw = ctypes.c_void_p()
    assert lib.accblas_comm_create(ctypes.byref(raw), 2, 2, dummy) == -2
    assert b"rank" in lib.accblas_comm_last_error_string()
    assert lib.accblas_comm_acc_dot(None, None, 1, 0, 1, 8, dummy, 1, dummy, 1, dummy,
                                    None) == -3
    assert lib.accblas_comm_acc_gemv(None, 2, 1, 4, 4, 1.0, dummy, 4, dummy, 1, 0.0,
                                     dummy, 1, None, None) == -3
    assert lib.accblas_comm_destroy(None) == 0


def test_shard_bounds_properties():
    """Property test of the partition both multi-GPU layers use: the pieces tile
    [0, n) in rank order, sizes differ by at most one (or by less than `align`
    at aligned interior edges), and the C and Python implementations agree."""
    from hypothesis import given, settings, strategies as st
    from accblas import comm, multi_gpu

    @settings(max_examples=200, deadline=None)
    @given(n=st.integers(0, 1 << 40), world=st.integers(1, 64),
           align=st.sampled_from([1, 2, 8, 64]))
    def check(n, world, align):
        pieces = [comm.shard_bounds(n, world, r, align) for r in range(world)]
        assert pieces[0][0] == 0 and pieces[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        assert all(lo <= hi for lo, hi in pieces)
        assert pieces == [multi_gpu.shard_bounds(n, world, r, align) for r in range(world)]
        if align == 1:
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1
        else:
            assert all(p[1] % align == 0 for p in pieces[:-1])

    check()


def test_no_environment_variable_redirects_the_binding(monkeypatch):
    """bench.py and the parity tests measure the in-tree build: ACCBLAS_LIB_PATH
    (honoured until round 3) must not point the binding anywhere else. The
    developer probes that compare builds set `accblas.LIB_PATH` themselves."""
    import importlib
    monkeypatch.setenv("ACCBLAS_LIB_PATH", "/nonexistent/libaccblas_hip.so")
    fresh = importlib.reload(accblas)
    try:
        assert fresh.LIB_PATH == os.path.join(fresh.PACKAGE_ROOT, "lib", "libaccblas_hip.so")
        path, sha = fresh.library_identity()
        assert path == fresh.LIB_PATH and len(sha) == 64
    finally:
        monkeypatch.delenv("ACCBLAS_LIB_PATH")
        importlib.reload(accblas)
