/*
 * accblas_vendor.h — C ABI of the vendor-library columns and setup helpers
 * (libaccblas_vendor.so, links rocBLAS + rocSOLVER). Separate from accblas.h on
 * purpose: nothing here is on the hot path, and the hot-path library must not
 * depend on vendor BLAS.
 *
 * Replaces the reference's cuBLAS/cuSOLVER call sites (file:line relative to
 * /root/reference): `cublas_gemv` cuda/gemv_kernels.cuh:198-243, `cublas_dot`
 * cuda/dot_kernels.cuh:268-299 (device pointer mode, cuda/dot_benchmark.cu:78-79),
 * `cublas_trsv` cuda/trsv_kernels.cuh:964-1008, and the in-place
 * `cusolverDnDgetrf` of cuda/trsv_memory.cuh:131-168.
 * Same conventions as accblas.h (int status, int64 sizes, opaque device
 * pointers); only f32 and f64 are supported.
 */
#ifndef ACCBLAS_VENDOR_H_
#define ACCBLAS_VENDOR_H_

#include "accblas.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct accblas_vendor_handle accblas_vendor_handle;

/* `cublas_get_handle` (cuda/utils.cuh:189-197) */
int accblas_vendor_create(accblas_vendor_handle **handle);
int accblas_vendor_destroy(accblas_vendor_handle *handle);

/* res = alpha*A*x + beta*res for ROW-major A through rocblas_{s,d}gemv with
 * operation_transpose on the column-major view (cuda/gemv_kernels.cuh:236-242) */
int accblas_vendor_gemv(accblas_vendor_handle *handle, accblas_dtype vt,
                        int64_t rows, int64_t cols, double alpha, const void *A,
                        int64_t lda, const void *x, int64_t incx, double beta,
                        void *res, int64_t incres);

/* *res (DEVICE scalar) = x . y through rocblas_{s,d}dot in device pointer mode */
int accblas_vendor_dot(accblas_vendor_handle *handle, accblas_dtype vt,
                       int64_t n, const void *x, int64_t incx, const void *y,
                       int64_t incy, void *res);

/* x <- T^{-1} x for ROW-major A: fill mode swapped + operation_transpose
 * (cuda/trsv_kernels.cuh:999-1007) */
int accblas_vendor_trsv(accblas_vendor_handle *handle, accblas_dtype vt,
                        accblas_uplo uplo, accblas_diag diag, int64_t n,
                        const void *A, int64_t lda, void *x, int64_t incx);

/* In-place partial-pivot LU (rocsolver_dgetrf_64: n * lda may exceed 2^31
 * elements, as it does for n = 65536) of the n x n fp64 device buffer
 * interpreted COLUMN-major with leading dimension lda, pivots discarded, as
 * cuda/trsv_memory.cuh:150-166 does. Unlike the reference, `info` (host int,
 * may be NULL) is returned so the caller can check it. */
int accblas_vendor_getrf(accblas_vendor_handle *handle, int64_t n, void *A,
                         int64_t lda, int *info);

#ifdef __cplusplus
}
#endif

#endif /* ACCBLAS_VENDOR_H_ */
