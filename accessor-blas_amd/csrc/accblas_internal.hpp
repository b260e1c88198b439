// Shared plumbing of the C-ABI translation units: error reporting, dtype
// dispatch, the handle layout. Nothing here is visible through include/accblas.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/accblas.h"
#include "dot_kernels.hpp"


struct accblas_handle {
    int device;
    int num_cus;
    accblas_dot_mode dot_mode;
    // one device allocation: [dot slots | scalar(16 B) | trsv helper(16 B) |
    // dot status(16 B)]
    void *workspace;
    accblas::dev::dot_workspace dot_ws;
    std::uint32_t *trsv_helper;
};


namespace accblas {


/** Records `message` as the calling thread's last error and returns `code`. */
int set_error(int code, const std::string &message);

/** hipError_t -> status; records file-free message "what: hipGetErrorString". */
int check_hip(hipError_t err, const char *what);


template <typename T>
struct type_tag {
    using type = T;
};

using f16 = _Float16;


/** Calls f(type_tag<T>) for the element type named by `t`. */
template <typename F>
int dispatch_type(accblas_dtype t, F &&f)
{
    switch (t) {
    case ACCBLAS_F16:
        return f(type_tag<f16>{});
    case ACCBLAS_F32:
        return f(type_tag<float>{});
    case ACCBLAS_F64:
        return f(type_tag<double>{});
    case ACCBLAS_I16:
    case ACCBLAS_I32:
        return set_error(ACCBLAS_ERR_UNSUPPORTED_TYPES,
                         "integer storage: accblas_scaled_* entry points only");
    }
    return set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "unknown accblas_dtype");
}

/** Arithmetic types are fp32 and fp64 only. */
template <typename F>
int dispatch_arithmetic(accblas_dtype t, F &&f)
{
    switch (t) {
    case ACCBLAS_F32:
        return f(type_tag<float>{});
    case ACCBLAS_F64:
        return f(type_tag<double>{});
    default:
        return set_error(ACCBLAS_ERR_UNSUPPORTED_TYPES,
                         "arithmetic type must be f32 or f64");
    }
}

/** Argument checks shared by the GEMV / dot entry points (before any HIP call). */
inline int validate_gemv(int64_t rows, int64_t cols, const void *A, int64_t lda,
                  const void *x, int64_t incx, const void *res, int64_t incres)
{
    if (rows < 0 || cols < 0) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "gemv: negative size");
    }
    if (lda < cols) {
        // `gen_mtx` rejects the same condition (cuda/matrix_helper.cuh:31-33)
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "gemv: row stride smaller than cols");
    }
    if (incx < 1 || incres < 1) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "gemv: increments must be >= 1");
    }
    if (rows > 0 && res == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "gemv: res is NULL");
    }
    if (rows > 0 && cols > 0 && (A == nullptr || x == nullptr)) {
        return set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "gemv: A or x is NULL");
    }
    return ACCBLAS_OK;
}

inline int validate_dot(const accblas_handle *handle, int64_t n, const void *x,
                 int64_t incx, const void *y, int64_t incy, const void *res)
{
    if (handle == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "dot: handle is NULL");
    }
    if (n < 0) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "dot: n < 0");
    }
    if (incx < 1 || incy < 1) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "dot: increments must be >= 1");
    }
    if (res == nullptr) {
        return set_error(ACCBLAS_ERR_NULL_POINTER, "dot: res is NULL");
    }
    if (n > 0 && (x == nullptr || y == nullptr)) {
        return set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "dot: x or y is NULL");
    }
    return ACCBLAS_OK;
}


/** Option "gemv_skew" (accblas_set_option; defined in accblas_trsv.hip next
 *  to the other options): 0 never, 1 automatic (default), 2 always. */
int gemv_skew_option_value();
/** Option "gemv_vector_rows": 0 = 16-byte loads only for 16-byte aligned
 *  operands (the rule until round 3), 1 = for every operand (default). */
int gemv_vector_rows_option_value();

inline hipStream_t as_stream(void *stream)
{
    return static_cast<hipStream_t>(stream);
}


}  // namespace accblas
