// Shared definitions of the accessor headers (namespace gko::acc).
//
// The reference does not ship these headers: its CMake clones Ginkgo
// (`/root/reference/CMakeLists.txt:18-33`) and includes <accessor/range.hpp>
// and <accessor/reduced_row_major.hpp> (`cuda/gemv_kernels.cuh:11-12`). This
// is a from-scratch implementation of the API surface those call sites use
// (SURVEY.md §8a row A0), usable from g++ (host) and hipcc (host + gfx950).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <type_traits>

// hipcc compiles every accessor member for both the host and gfx950; a plain
// host compiler (g++ -std=c++14) sees ordinary inline functions.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GKO_ACC_ATTRIBUTES __host__ __device__
#define GKO_ACC_INLINE __forceinline__
#else
#define GKO_ACC_ATTRIBUTES
#define GKO_ACC_INLINE inline
#endif


namespace gko {
namespace acc {


/**
 * Index / length type of every accessor. It has to be a signed 64 bit integer:
 * the reference hands `matrix_info::size` (a std::array<std::int64_t, 2>,
 * `cuda/utils.cuh:19-21`) directly to the range constructor
 * (`cuda/gemv_kernels.cuh:187`).
 */
using size_type = std::int64_t;


namespace detail {


template <bool... Values>
struct all_true;

template <>
struct all_true<> : std::true_type {};

template <bool First, bool... Rest>
struct all_true<First, Rest...>
    : std::integral_constant<bool, First && all_true<Rest...>::value> {};


template <typename... Args>
struct are_all_integral
    : all_true<std::is_integral<std::decay_t<Args>>::value...> {};


/** Plain fixed-size array that is trivially usable in device code. */
template <typename T, std::size_t N>
struct fixed_array {
    T data[N > 0 ? N : 1];

    GKO_ACC_ATTRIBUTES constexpr const T &operator[](std::size_t i) const
    {
        return data[i];
    }
    GKO_ACC_ATTRIBUTES constexpr T &operator[](std::size_t i)
    {
        return data[i];
    }
};


template <typename T, std::size_t N, std::size_t... Idx>
constexpr fixed_array<T, N> to_fixed_impl(const std::array<T, N> &in,
                                          std::index_sequence<Idx...>)
{
    return fixed_array<T, N>{{in[Idx]...}};
}

template <typename T, std::size_t N>
constexpr fixed_array<T, N> to_fixed(const std::array<T, N> &in)
{
    return to_fixed_impl(in, std::make_index_sequence<N>{});
}


template <typename T, std::size_t N, std::size_t... Idx>
constexpr std::array<T, N> to_std_impl(const fixed_array<T, N> &in,
                                       std::index_sequence<Idx...>)
{
    return std::array<T, N>{{in[Idx]...}};
}

template <typename T, std::size_t N>
constexpr std::array<T, N> to_std(const fixed_array<T, N> &in)
{
    return to_std_impl(in, std::make_index_sequence<N>{});
}


}  // namespace detail
}  // namespace acc
}  // namespace gko
