// Host launchers of the DOT variants, C++14 (g++). Same surface as
// `/root/reference/cuda/dot_kernels.cuh`: `myBlasHandle` (:29-65), `dot<V>`
// (:192-206), `acc_dot<ArType, StType, ResType>` (:224-263) and the vendor
// column (`cublas_dot` :291-299, here `rocblas_dot`). The result is written to
// a DEVICE scalar, as in the reference.
#pragma once

#include <array>
#include <cinttypes>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>

#include <accblas.h>
#include <accblas_vendor.h>

#include "utils.hpp"


/**
 * RAII wrapper of what every hand-written BLAS-1 launch needs: the device's
 * compute-unit count (sizes the persistent grid) and a little device scratch.
 */
class myBlasHandle {
public:
    myBlasHandle() { HIP_CALL(accblas_handle_create(&handle_)); }
    ~myBlasHandle() { accblas_handle_destroy(handle_); }
    myBlasHandle(const myBlasHandle &) = delete;
    myBlasHandle &operator=(const myBlasHandle &) = delete;

    /** Number of compute units of the current device. */
    int get_num_compute_units() const
    {
        int cus{};
        HIP_CALL(accblas_handle_num_cus(handle_, &cus));
        return cus;
    }

    /** Device pointer to one value of type T (16 bytes of scratch). */
    template <typename T>
    T *get_device_value_ptr()
    {
        static_assert(sizeof(T) <= 16, "type too large for the device storage");
        void *ptr{};
        HIP_CALL(accblas_handle_device_scratch(handle_, &ptr));
        return static_cast<T *>(ptr);
    }

    /** Deterministic (default) or reference-like atomic combination. */
    void set_dot_mode(accblas_dot_mode mode)
    {
        HIP_CALL(accblas_handle_set_dot_mode(handle_, mode));
    }

    accblas_handle *get() { return handle_; }

private:
    accblas_handle *handle_{};
};


/** *res = sum_i x[i * x_info.stride] * y[i * y_info.stride], all ValueType. */
template <typename ValueType>
void dot(myBlasHandle *handle, const matrix_info x_info, const ValueType *x,
         const matrix_info y_info, const ValueType *y, ValueType *res)
{
    HIP_CALL(accblas_dot(handle->get(), accblas_type_of<ValueType>::value,
                         x_info.size[0], x, x_info.stride, y, y_info.stride,
                         res, nullptr));
}


/** Same through the accessor: StType loads, ArType accumulation, result
 *  written as ResType. */
template <typename ArType, typename StType, typename ResType>
void acc_dot(myBlasHandle *handle, const matrix_info x_info, const StType *x,
             const matrix_info y_info, const StType *y, ResType *res)
{
    // Accessor setup
    constexpr std::size_t dimensionality{2};
    std::array<gko::acc::size_type, dimensionality - 1> x_stride{
        {x_info.stride}};
    std::array<gko::acc::size_type, dimensionality - 1> y_stride{
        {y_info.stride}};

    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    auto x_acc = c_range(x_info.size, x, x_stride);
    auto y_acc = c_range(y_info.size, y, y_stride);

    HIP_CALL(accblas_acc_dot(
        handle->get(), accblas_type_of<ArType>::value,
        accblas_type_of<StType>::value, accblas_type_of<ResType>::value,
        x_acc.length(0), x_acc->get_const_storage(), x_acc->get_stride()[0],
        y_acc->get_const_storage(), y_acc->get_stride()[0], res, nullptr));
}


/** Vendor column: rocBLAS dot, result to a device scalar. */
template <typename ValueType>
void rocblas_dot(accblas_vendor_handle *handle, const matrix_info x_info,
                 const ValueType *x, const matrix_info y_info,
                 const ValueType *y, ValueType *res)
{
    HIP_CALL(accblas_vendor_dot(handle, accblas_type_of<ValueType>::value,
                                x_info.size[0], x, x_info.stride, y,
                                y_info.stride, res));
}
