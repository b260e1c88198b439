// Host launchers of the GEMV variants, C++14 (g++). Same templates as the
// bottom half of `/root/reference/cuda/gemv_kernels.cuh`: `gemv<ValueType>`
// (:136-147), `acc_gemv<ArType, StType>` (:168-193) and the vendor column
// (`cublas_gemv` :230-243, here `rocblas_gemv`). The kernels live behind the C
// ABI (include/accblas.h) in libaccblas_hip.so; the accessor objects are built
// here exactly as the reference builds them and unpacked into the C call.
#pragma once

#include <array>
#include <cinttypes>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>

#include <accblas.h>
#include <accblas_vendor.h>

#include "utils.hpp"


/**
 * res = alpha * mtx * x + beta * res with the plain-pointer kernel.
 * mtx is row-major, m_info.stride its row stride; x_info.stride and
 * res_info.stride are the vector increments.
 */
template <typename ValueType>
void gemv(const matrix_info m_info, ValueType alpha, const ValueType *mtx,
          const matrix_info x_info, const ValueType *x,
          const matrix_info res_info, ValueType beta, ValueType *res)
{
    HIP_CALL(accblas_gemv(accblas_type_of<ValueType>::value, m_info.size[0],
                          m_info.size[1], static_cast<double>(alpha), mtx,
                          m_info.stride, x, x_info.stride,
                          static_cast<double>(beta), res, res_info.stride,
                          nullptr));
}


/**
 * res = alpha * mtx * x + beta * res with the kernel driven through
 * range<reduced_row_major<2, ArType, StType>>: StType in memory, ArType in
 * registers.
 */
template <typename ArType, typename StType>
void acc_gemv(const matrix_info m_info, ArType alpha, const StType *mtx,
              const matrix_info x_info, const StType *x,
              const matrix_info res_info, ArType beta, StType *res)
{
    // Accessor setup
    constexpr std::size_t dimensionality{2};
    std::array<gko::acc::size_type, dimensionality - 1> m_stride{
        {m_info.stride}};
    std::array<gko::acc::size_type, dimensionality - 1> x_stride{
        {x_info.stride}};
    std::array<gko::acc::size_type, dimensionality - 1> res_stride{
        {res_info.stride}};

    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    auto m_acc = c_range(m_info.size, mtx, m_stride);
    auto x_acc = c_range(x_info.size, x, x_stride);
    auto res_acc = range(res_info.size, res, res_stride);

    // The device side re-creates the same three ranges from these fields
    // (csrc/accblas_gemv.hip) and launches the accessor kernel with them.
    HIP_CALL(accblas_acc_gemv(
        accblas_type_of<ArType>::value, accblas_type_of<StType>::value,
        m_acc.length(0), m_acc.length(1), static_cast<double>(alpha),
        m_acc->get_const_storage(), m_acc->get_stride()[0],
        x_acc->get_const_storage(), x_acc->get_stride()[0],
        static_cast<double>(beta), res_acc->get_stored_data(),
        res_acc->get_stride()[0], nullptr));
}


/** Vendor column: rocBLAS gemv on the row-major matrix. */
template <typename ValueType>
void rocblas_gemv(accblas_vendor_handle *handle, const matrix_info m_info,
                  ValueType alpha, const ValueType *mtx,
                  const matrix_info x_info, const ValueType *x,
                  const matrix_info res_info, ValueType beta, ValueType *y)
{
    HIP_CALL(accblas_vendor_gemv(
        handle, accblas_type_of<ValueType>::value, m_info.size[0],
        m_info.size[1], static_cast<double>(alpha), mtx, m_info.stride, x,
        x_info.stride, static_cast<double>(beta), y, res_info.stride));
}
