// Device-side views the kernels are written against.
//
// The reference has two kernel families per operation: plain pointers plus
// `matrix_info` (`cuda/gemv_kernels.cuh:30-64`) and accessor ranges
// (`:79-113`). Here ONE kernel template serves both: it is instantiated
// either with `gko::acc::range<reduced_row_major<2, Ar, St>>` arguments
// (the accessor path) or with `plain_view<V>` (pointer + sizes + stride, the
// plain path). Both expose `length(d)` and `operator()(i, j)`; `view_traits`
// adds what the 16-byte vector fast path needs (storage pointer + row
// stride). Identical instruction order in both instantiations is what makes
// "plain == Acc<T,T> bit for bit"  hold by construction.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>
#include <accessor/row_major.hpp>


namespace accblas {
namespace dev {


using index_type = std::int64_t;


/** Row-major 2-D view over a plain pointer: element (i, j) = data[i*stride+j].
 *  A vector with increment inc is the (n, 1) view with stride = inc, exactly
 *  like the reference's `x[col * x_info.stride]` (`gemv_kernels.cuh:51`). */
template <typename ValueType>
struct plain_view {
    using storage_type = ValueType;
    using arithmetic_type = std::remove_cv_t<ValueType>;

    ValueType *data;
    index_type size[2];
    index_type stride0;

    __host__ __device__ __forceinline__ index_type length(index_type d) const
    {
        return d < 2 ? size[d] : 1;
    }

    __host__ __device__ __forceinline__ ValueType &operator()(
        index_type i, index_type j) const
    {
        return data[i * stride0 + j];
    }
};


/**
 * What the vector fast paths need from a view. `is_pointer_view` says whether
 * element (i, j) IS storage[i * stride + j] converted with a static_cast, i.e.
 * whether 16-byte loads of the storage type may replace `operator()`. It holds
 * for plain pointers, `reduced_row_major` and `row_major`; any other accessor
 * kind (e.g. `scaled_reduced_row_major`, whose values are storage * scalar)
 * gets the fallback below and runs through the kernels' `operator()` paths
 * ("all others work accordingly", `/root/reference/README.md:19`).
 */
template <typename View>
struct view_traits;

template <typename Accessor>
struct view_traits<gko::acc::range<Accessor>> {
    static constexpr bool is_pointer_view = false;
    using storage_type = typename Accessor::storage_type;
    using arithmetic_type = typename Accessor::arithmetic_type;

    __host__ __device__ static __forceinline__ storage_type *data(
        const gko::acc::range<Accessor> &)
    {
        return nullptr;
    }
    __host__ __device__ static __forceinline__ index_type stride(
        const gko::acc::range<Accessor> &)
    {
        return 0;
    }
};

template <typename ValueType>
struct view_traits<plain_view<ValueType>> {
    static constexpr bool is_pointer_view = true;
    using storage_type = ValueType;
    using arithmetic_type = std::remove_cv_t<ValueType>;

    __host__ __device__ static __forceinline__ ValueType *data(
        const plain_view<ValueType> &v)
    {
        return v.data;
    }
    __host__ __device__ static __forceinline__ index_type stride(
        const plain_view<ValueType> &v)
    {
        return v.stride0;
    }
};

template <typename ArType, typename StType>
struct view_traits<
    gko::acc::range<gko::acc::reduced_row_major<2, ArType, StType>>> {
    static constexpr bool is_pointer_view = true;
    using range_type =
        gko::acc::range<gko::acc::reduced_row_major<2, ArType, StType>>;
    using storage_type = StType;
    using arithmetic_type = std::remove_cv_t<ArType>;

    __host__ __device__ static __forceinline__ StType *data(
        const range_type &r)
    {
        return r->get_stored_data();
    }
    __host__ __device__ static __forceinline__ index_type stride(
        const range_type &r)
    {
        return r->stride(0);
    }
};


template <typename ValueType>
struct view_traits<gko::acc::range<gko::acc::row_major<ValueType, 2>>> {
    static constexpr bool is_pointer_view = true;
    using range_type = gko::acc::range<gko::acc::row_major<ValueType, 2>>;
    using storage_type = ValueType;
    using arithmetic_type = std::remove_cv_t<ValueType>;

    __host__ __device__ static __forceinline__ ValueType *data(
        const range_type &r)
    {
        return r->get_stored_data();
    }
    __host__ __device__ static __forceinline__ index_type stride(
        const range_type &r)
    {
        return r->stride(0);
    }
};


inline bool is_aligned(const void *ptr, std::size_t bytes)
{
    return reinterpret_cast<std::uintptr_t>(ptr) % bytes == 0;
}

__host__ __device__ constexpr index_type ceildiv(index_type a, index_type b)
{
    return (a + b - 1) / b;
}


/** N consecutive storage elements moved by ONE vector memory instruction
 *  (16 bytes: 8 x fp16 / 4 x fp32 / 2 x fp64). */
template <typename T, int N>
struct packed {
    typedef T type __attribute__((ext_vector_type(N)));
    // the same vector where only its elements are known to be aligned
    typedef T loose_type
        __attribute__((ext_vector_type(N), aligned(sizeof(T))));
};

/** Streaming load of N storage elements by one instruction. `ptr` need only be
 *  aligned like an element (the queues run in the hardware's unaligned mode and
 *  the instruction is the same either way); a load that is not N*sizeof(T)
 *  aligned may lie across two cache lines and then costs two requests. Only
 *  GEMV's long-row shape (`RowAlign`: a pitch that is not a multiple of 128
 *  bytes, or rows that are not 16-byte aligned, with enough columns) peels to
 *  an aligned start; its other shapes, dot (chunks start at the operands'
 *  bases as they are) and the TRSV sweeps take the data as it comes. Intended:
 *  the bits of a result do not depend on the alignment of its operands' bases.
 *  NT = non-temporal hint (`global_load_dwordx4 ... nt`) for data that is
 *  read exactly once (the matrix in GEMV/TRSV, both vectors in dot). */
template <int N, bool NT, typename T>
__device__ __forceinline__ typename packed<std::remove_cv_t<T>, N>::type
load_packed(const T *ptr)
{
    using loose_type = typename packed<std::remove_cv_t<T>, N>::loose_type;
    const auto *vptr = reinterpret_cast<const loose_type *>(ptr);
    if (NT) {
        return __builtin_nontemporal_load(vptr);
    }
    return *vptr;
}


}  // namespace dev
}  // namespace accblas
