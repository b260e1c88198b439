// gko::acc::row_major<ValueType, Dim>: the plain row-major accessor (storage
// type == arithmetic type, operator() returns a real reference). Part of the
// accessor family the reference's README points to ("all others work
// accordingly", `/root/reference/README.md:19`); SURVEY.md §8f N4.
#pragma once

#include "index_span.hpp"
#include "range.hpp"
#include "utils.hpp"


namespace gko {
namespace acc {


template <typename ValueType, std::size_t Dimensionality>
class row_major {
public:
    friend class range<row_major>;

    static_assert(Dimensionality >= 1, "an accessor has at least 1 dimension");

    using value_type = ValueType;
    using arithmetic_type = std::remove_cv_t<ValueType>;
    using storage_type = ValueType;
    using data_type = value_type *;
    using const_accessor = row_major<const ValueType, Dimensionality>;
    static constexpr std::size_t dimensionality = Dimensionality;
    using length_type = std::array<size_type, dimensionality>;
    using stride_type = std::array<size_type, dimensionality - 1>;

protected:
    static constexpr std::size_t num_strides = dimensionality - 1;

    constexpr row_major(length_type size, data_type data, stride_type stride)
        : lengths_(detail::to_fixed(size)),
          data_{data},
          stride_(detail::to_fixed(stride))
    {}

    template <typename... Strides,
              typename = std::enable_if_t<
                  (sizeof...(Strides) > 0) &&
                  sizeof...(Strides) == num_strides &&
                  detail::are_all_integral<Strides...>::value>>
    constexpr row_major(length_type size, data_type data, Strides... strides)
        : row_major{size, data,
                    stride_type{{static_cast<size_type>(strides)...}}}
    {}

    constexpr row_major(length_type size, data_type data)
        : lengths_(detail::to_fixed(size)),
          data_{data},
          stride_(dense_stride(size))
    {}

public:
    GKO_ACC_ATTRIBUTES constexpr range<const_accessor> to_const() const
    {
        return range<const_accessor>{detail::to_std(lengths_), data_,
                                     detail::to_std(stride_)};
    }

    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr std::enable_if_t<
        detail::are_all_integral<Indices...>::value &&
            sizeof...(Indices) == dimensionality,
        value_type &>
    operator()(Indices... indices) const
    {
        return data_[linear_index<0>(static_cast<size_type>(indices)...)];
    }

    template <typename... Spans>
    GKO_ACC_ATTRIBUTES constexpr std::enable_if_t<
        detail::are_all_spans<Spans...>::value &&
            sizeof...(Spans) == dimensionality,
        range<row_major>>
    operator()(Spans... spans) const
    {
        return range<row_major>{length_type{{spans.length()...}},
                                data_ + linear_index<0>(spans.begin...),
                                detail::to_std(stride_)};
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type length(
        size_type dimension) const
    {
        return dimension < static_cast<size_type>(dimensionality)
                   ? lengths_[static_cast<std::size_t>(dimension)]
                   : 1;
    }

    constexpr stride_type get_stride() const { return detail::to_std(stride_); }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type stride(
        std::size_t d) const
    {
        return stride_[d];
    }

    constexpr length_type get_size() const { return detail::to_std(lengths_); }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr data_type get_stored_data()
        const
    {
        return data_;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr const value_type *
    get_const_storage() const
    {
        return data_;
    }

private:
    static constexpr detail::fixed_array<size_type, num_strides> dense_stride(
        const length_type &size)
    {
        detail::fixed_array<size_type, num_strides> stride{};
        size_type running = 1;
        for (std::size_t d = num_strides; d > 0; --d) {
            running *= size[d];
            stride[d - 1] = running;
        }
        return stride;
    }

    template <std::size_t Dim>
    GKO_ACC_ATTRIBUTES constexpr size_type linear_index(size_type last) const
    {
        static_assert(Dim == dimensionality - 1, "wrong number of indices");
        return last;
    }

    template <std::size_t Dim, typename... Rest>
    GKO_ACC_ATTRIBUTES constexpr size_type linear_index(size_type first,
                                                        Rest... rest) const
    {
        return first * stride_[Dim] + linear_index<Dim + 1>(rest...);
    }

    detail::fixed_array<size_type, dimensionality> lengths_;
    data_type data_;
    detail::fixed_array<size_type, num_strides> stride_;
};


template <typename V, std::size_t D>
constexpr std::size_t row_major<V, D>::dimensionality;


}  // namespace acc
}  // namespace gko
