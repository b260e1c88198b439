// Half-open index interval [begin, end) used to cut sub-ranges out of a range.
// Part of the accessor API recalled from upstream Ginkgo (SURVEY.md §8a row A0,
// "sub-ranges via index spans"); the benchmarks address the top-left n×n block
// of the big matrix through size + stride instead
// (`/root/reference/cuda/gemv_benchmark.cu:215`).
#pragma once

#include "utils.hpp"


namespace gko {
namespace acc {


struct index_span {
    GKO_ACC_ATTRIBUTES constexpr index_span(size_type point) noexcept
        : begin{point}, end{point + 1}
    {}

    GKO_ACC_ATTRIBUTES constexpr index_span(size_type begin_,
                                            size_type end_) noexcept
        : begin{begin_}, end{end_}
    {}

    GKO_ACC_ATTRIBUTES constexpr bool is_valid() const { return begin < end; }

    GKO_ACC_ATTRIBUTES constexpr size_type length() const
    {
        return end - begin;
    }

    const size_type begin;
    const size_type end;
};


GKO_ACC_ATTRIBUTES constexpr bool operator==(const index_span &a,
                                             const index_span &b)
{
    return a.begin == b.begin && a.end == b.end;
}

GKO_ACC_ATTRIBUTES constexpr bool operator!=(const index_span &a,
                                             const index_span &b)
{
    return !(a == b);
}


namespace detail {


template <typename... Args>
struct are_all_spans
    : all_true<std::is_same<std::decay_t<Args>, index_span>::value...> {};


}  // namespace detail
}  // namespace acc
}  // namespace gko
