"""the HIP kernels driven through the OTHER accessor kinds
(`row_major`, `scaled_reduced_row_major`): a small HIP program compiled against
the kernel headers (every kind, TRSV included), and - since round 5 - the C-ABI
entry points `accblas_scaled_acc_gemv` / `accblas_scaled_acc_dot` (the
reference's own launchers only build `reduced_row_major` ranges)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(REPO, "accessor-blas_amd")


def test_kernels_accept_other_accessor_kinds(tmp_path):
    exe = tmp_path / "device_accessor_kinds"
    subprocess.run(
        ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "--offload-arch=gfx950",
         "-ffp-contract=fast", "-I", os.path.join(PKG, "include"), "-I",
         os.path.join(PKG, "csrc"), os.path.join(REPO, "tests", "cpp",
                                                 "device_accessor_kinds.hip"),
         "-o", str(exe)], check=True, timeout=900)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device accessor kinds ok" in out.stdout
    assert out.stdout.count(" ok\n") >= 6


# ---- the C-ABI entry points for scaled_reduced_row_major operands --------------
# (`accblas_scaled_acc_gemv` / `accblas_scaled_acc_dot`: the reference's README
# names the accessor family, `/root/reference/README.md:19`; its launchers only
# build reduced_row_major). Checked against numpy in extended precision on the
# same storage: value = storage x scalar, products and sums in long double.
SCALED_STORAGE = {"i16": (3, np.int16), "i32": (4, np.int32), "f16": (0, np.float16),
                  "f32": (1, np.float32)}
VECTOR_STORAGE = {"f32": (1, np.float32), "f64": (2, np.float64)}
ARITHMETIC = {"f32": (1, np.float32), "f64": (2, np.float64)}


def _scaled_matrix(rng, rows, lda, kind):
    if kind == "i16":
        return rng.integers(-32000, 32001, rows * lda).astype(np.int16), 1.0 / 32000.0
    if kind == "i32":
        return rng.integers(-(1 << 20), (1 << 20) + 1, rows * lda).astype(np.int32), 1.0 / (1 << 20)
    return rng.uniform(-1.0, 1.0, rows * lda).astype(SCALED_STORAGE[kind][1]), 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("ar_name", ["f64", "f32"])
@pytest.mark.parametrize("a_name", ["i16", "i32", "f16", "f32"])
@pytest.mark.parametrize("v_name", ["f32", "f64"])
@pytest.mark.parametrize("rows,cols,lda,incx,incres", [(257, 700, 701, 1, 1), (64, 33, 40, 2, 3),
                                                       (1, 1, 1, 1, 1), (300, 1025, 1025, 1, 1)])
def test_scaled_acc_gemv_through_the_c_abi(ar_name, a_name, v_name, rows, cols, lda, incx,
                                           incres):
    import torch
    import accblas
    ar, ar_np = ARITHMETIC[ar_name]
    a_st, _ = SCALED_STORAGE[a_name]
    v_st, v_np = VECTOR_STORAGE[v_name]
    rng = np.random.default_rng(rows * 1000 + cols)
    A, unit = _scaled_matrix(rng, rows, lda, a_name)
    scalars = ((1.0 + 0.5 * rng.uniform(-1.0, 1.0, cols)) * unit).astype(ar_np)
    x = np.full(cols * incx, 7.0, dtype=v_np)
    x[::incx] = rng.uniform(-1.0, 1.0, cols).astype(v_np)
    res = np.full(rows * incres, -3.0, dtype=v_np)
    res[::incres] = rng.uniform(-1.0, 1.0, rows).astype(v_np)
    alpha, beta = 1.5, 0.25
    values = A.reshape(rows, lda)[:, :cols].astype(np.longdouble) * scalars.astype(np.longdouble)
    want = (np.longdouble(ar_np(alpha)) * (values @ x[::incx].astype(np.longdouble)) +
            np.longdouble(ar_np(beta)) * res[::incres].astype(np.longdouble))
    dA, ds, dx, dres = (torch.from_numpy(v).cuda() for v in (A, scalars, x, res))
    accblas.scaled_acc_gemv(ar, a_st, v_st, rows, cols, alpha, dA, lda, ds, dx, incx, beta, dres,
                            incres)
    torch.cuda.synchronize()
    got = dres.cpu().numpy()
    assert np.all(got.reshape(rows, incres)[:, 1:] == -3.0)  # between the increments: untouched
    err = float(np.abs(want - got[::incres].astype(np.longdouble)).sum() /
                max(float(np.abs(want).sum()), 1e-300))
    # (arithmetic eps x the row length's square root for the sum, plus the
    # rounding of the result into its storage type)
    tol = {"f64": 1e-13, "f32": 5e-6}[ar_name] + {"f64": 0.0, "f32": 1e-7}[v_name]
    assert err <= tol, err


@pytest.mark.gpu
@pytest.mark.parametrize("ar_name", ["f64", "f32"])
@pytest.mark.parametrize("x_name", ["i16", "i32", "f16", "f32"])
@pytest.mark.parametrize("y_name", ["f32", "f64"])
@pytest.mark.parametrize("n,incx,incy", [(65537, 1, 1), (1000, 3, 2), (1, 1, 1), (0, 1, 1)])
def test_scaled_acc_dot_through_the_c_abi(ar_name, x_name, y_name, n, incx, incy):
    import torch
    import accblas
    ar, ar_np = ARITHMETIC[ar_name]
    x_st, _ = SCALED_STORAGE[x_name]
    y_st, y_np = VECTOR_STORAGE[y_name]
    rng = np.random.default_rng(n + 17)
    x, unit = _scaled_matrix(rng, max(n, 1), incx, x_name)
    scale = np.array([1.25 * unit], dtype=ar_np)
    y = rng.uniform(-1.0, 1.0, max(n, 1) * incy).astype(y_np)
    terms = (x[::incx][:n].astype(np.longdouble) * np.longdouble(scale[0]) *
             y[::incy][:n].astype(np.longdouble))
    want = terms.sum() if n else np.longdouble(0)
    spread = float(np.sqrt((terms * terms).sum())) if n else 0.0
    handle = accblas.Handle()
    try:
        dres = torch.full((1,), 5.0, dtype=accblas.torch_dtype(ar), device="cuda")
        accblas.scaled_acc_dot(handle, ar, x_st, y_st, n, torch.from_numpy(x).cuda(), incx,
                               torch.from_numpy(scale).cuda(), torch.from_numpy(y).cuda(), incy,
                               dres)
        torch.cuda.synchronize()
        assert not accblas.dot_status(handle)
    finally:
        handle.close()
    eps = {"f64": 2.0 ** -53, "f32": 2.0 ** -24}[ar_name]
    assert abs(float(dres.item()) - float(want)) <= 64 * eps * spread + 4 * eps * abs(float(want))


@pytest.mark.gpu
def test_scaled_entry_points_validate_their_types():
    import torch
    import accblas
    lib = accblas.load()
    buf = torch.zeros(64, dtype=torch.float64, device="cuda")
    p = buf.data_ptr()
    handle = accblas.Handle()
    try:
        # fp64 as SCALED storage, fp16 vectors, integer arithmetic: unsupported pairs
        assert lib.accblas_scaled_acc_gemv(2, 2, 1, 4, 4, 1.0, p, 4, p, p, 1, 0.0, p, 1, None) == -1
        assert lib.accblas_scaled_acc_gemv(2, 3, 0, 4, 4, 1.0, p, 4, p, p, 1, 0.0, p, 1, None) == -1
        assert lib.accblas_scaled_acc_gemv(3, 3, 1, 4, 4, 1.0, p, 4, p, p, 1, 0.0, p, 1, None) == -1
        assert lib.accblas_scaled_acc_dot(handle.raw, 2, 2, 1, 4, p, 1, p, p, 1, p, None) == -1
        # the scalars are required, the usual shape checks apply
        assert lib.accblas_scaled_acc_gemv(2, 3, 1, 4, 4, 1.0, p, 4, None, p, 1, 0.0, p, 1, None) == -3
        assert lib.accblas_scaled_acc_gemv(2, 3, 1, 4, 8, 1.0, p, 4, p, p, 1, 0.0, p, 1, None) == -2
        assert lib.accblas_scaled_acc_dot(handle.raw, 2, 3, 1, 4, p, 1, None, p, 1, p, None) == -3
        # integer storage stays out of the reference's own launchers
        assert lib.accblas_acc_gemv(2, 3, 4, 4, 1.0, p, 4, p, 1, 0.0, p, 1, None) == -1
        assert lib.accblas_dtype_size(3) == 2 and lib.accblas_dtype_size(4) == 4
    finally:
        handle.close()
