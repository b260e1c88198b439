// Host launchers of the TRSV variants, C++14 (g++). Same surface as
// `/root/reference/cuda/trsv_kernels.cuh`: `tmtx_t` (:22), `dmtx_t` (:29),
// `trsv<V>` (:455-488), `acc_trsv<ArType, StType>` (:918-961) and the vendor
// column (`cublas_trsv` :994-1008, here `rocblas_trsv`). x is both right-hand
// side and solution; `trsv_helper` are two caller-owned device words.
#pragma once

#include <array>
#include <cinttypes>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>

#include <accblas.h>
#include <accblas_vendor.h>

#include "utils.hpp"


/** Which triangle of the row-major matrix holds the system. */
enum class tmtx_t { upper, lower };

/** Whether the diagonal is read (non_unit) or taken as all ones (unit). */
enum class dmtx_t { non_unit, unit };


inline accblas_uplo to_accblas(tmtx_t t)
{
    return t == tmtx_t::upper ? ACCBLAS_UPPER : ACCBLAS_LOWER;
}

inline accblas_diag to_accblas(dmtx_t d)
{
    return d == dmtx_t::unit ? ACCBLAS_UNIT : ACCBLAS_NON_UNIT;
}


template <typename ValueType>
void trsv(const matrix_info m_info, tmtx_t ttype, dmtx_t dtype,
          const ValueType *mtx, const matrix_info x_info, ValueType *x,
          std::uint32_t *trsv_helper)
{
    HIP_CALL(accblas_trsv(accblas_type_of<ValueType>::value, to_accblas(ttype),
                          to_accblas(dtype), m_info.size[0], mtx, m_info.stride,
                          x, x_info.stride, trsv_helper, nullptr));
}


template <typename ArType, typename StType>
void acc_trsv(const matrix_info m_info, tmtx_t ttype, dmtx_t dtype,
              const StType *mtx, const matrix_info x_info, StType *x,
              std::uint32_t *trsv_helper)
{
    // Accessor setup
    constexpr std::size_t dimensionality{2};
    std::array<gko::acc::size_type, dimensionality - 1> m_stride{
        {m_info.stride}};
    std::array<gko::acc::size_type, dimensionality - 1> x_stride{
        {x_info.stride}};

    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    auto m_acc = c_range(m_info.size, mtx, m_stride);
    auto x_acc = range(x_info.size, x, x_stride);

    HIP_CALL(accblas_acc_trsv(
        accblas_type_of<ArType>::value, accblas_type_of<StType>::value,
        to_accblas(ttype), to_accblas(dtype), m_acc.length(0),
        m_acc->get_const_storage(), m_acc->get_stride()[0],
        x_acc->get_stored_data(), x_acc->get_stride()[0], trsv_helper,
        nullptr));
}


/** Vendor column: rocBLAS trsv on the row-major matrix. */
template <typename ValueType>
void rocblas_trsv(accblas_vendor_handle *handle, tmtx_t ttype, dmtx_t dtype,
                  const matrix_info m_info, const ValueType *mtx,
                  const matrix_info x_info, ValueType *x)
{
    HIP_CALL(accblas_vendor_trsv(handle, accblas_type_of<ValueType>::value,
                                 to_accblas(ttype), to_accblas(dtype),
                                 m_info.size[0], mtx, m_info.stride, x,
                                 x_info.stride));
}
