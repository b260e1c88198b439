// gko::acc::range<Accessor>: the multi-dimensional view that kernels receive
// by value as a launch argument.
//
// Call sites this serves: `/root/reference/cuda/gemv_kernels.cuh:178-192`
// (construction from size / pointer / stride, passed to <<<>>> by value),
// `:93,101-110` (length(), operator()), `cuda/trsv_kernels.cuh:530` (deduced
// as `gko::acc::range<MtxAccessor>` in a kernel signature).
#pragma once

#include <utility>

#include "utils.hpp"


namespace gko {
namespace acc {


template <typename Accessor>
class range {
public:
    using accessor = Accessor;

    static constexpr std::size_t dimensionality = accessor::dimensionality;

    ~range() = default;
    range(const range &) = default;
    range(range &&) = default;

    /** Forwards everything to the accessor constructor, e.g.
     *  range(std::array<size_type, D> size, St *data,
     *        std::array<size_type, D - 1> stride). */
    template <typename First, typename... Rest,
              typename = std::enable_if_t<
                  !std::is_same<std::decay_t<First>, range>::value>>
    GKO_ACC_ATTRIBUTES constexpr explicit range(First &&first, Rest &&... rest)
        : accessor_{std::forward<First>(first), std::forward<Rest>(rest)...}
    {}

    /** Element access (integral indices) or sub-range (index_span). */
    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr auto operator()(
        Indices &&... indices) const
        -> decltype(std::declval<const accessor &>()(
            std::forward<Indices>(indices)...))
    {
        return accessor_(std::forward<Indices>(indices)...);
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type length(
        size_type dimension) const
    {
        return accessor_.length(dimension);
    }

    GKO_ACC_ATTRIBUTES constexpr const accessor *operator->() const noexcept
    {
        return &accessor_;
    }

    GKO_ACC_ATTRIBUTES constexpr const accessor &get_accessor() const noexcept
    {
        return accessor_;
    }

private:
    accessor accessor_;
};


template <typename Accessor>
constexpr std::size_t range<Accessor>::dimensionality;


}  // namespace acc
}  // namespace gko
