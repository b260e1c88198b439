// gko::acc::reduced_row_major<Dim, ArithmeticType, StorageType>: row-major
// accessor that stores values as StorageType and hands them out as
// ArithmeticType, i.e. storage precision decoupled from arithmetic precision.
//
// Call sites this serves (the reference relies on Ginkgo's header, which is
// not in its tree): `/root/reference/cuda/gemv_kernels.cuh:178-192,86-110`,
// `cuda/dot_kernels.cuh:233-256,138,153-154`,
// `cuda/trsv_kernels.cuh:923-958,541,574-576,641,651,672-673,683`.
// Semantics: element (i0, ..., i_{D-1}) lives at
// storage[i0*stride[0] + ... + i_{D-2}*stride[D-2] + i_{D-1}]; a read is
// static_cast<Ar>(stored), a write is static_cast<St>(value).
#pragma once

#include "index_span.hpp"
#include "range.hpp"
#include "reference_helper.hpp"
#include "utils.hpp"


namespace gko {
namespace acc {


template <std::size_t Dimensionality, typename ArithmeticType,
          typename StorageType>
class reduced_row_major {
public:
    using arithmetic_type = std::remove_cv_t<ArithmeticType>;
    using storage_type = StorageType;
    static constexpr std::size_t dimensionality = Dimensionality;
    static constexpr bool is_const = std::is_const<storage_type>::value;

    using const_accessor =
        reduced_row_major<dimensionality, arithmetic_type, const storage_type>;
    using length_type = std::array<size_type, dimensionality>;
    using storage_stride_type =
        std::array<size_type, (dimensionality > 0 ? dimensionality - 1 : 0)>;
    using reference_type =
        reference_class::reduced_storage<arithmetic_type, storage_type>;
    /** What operator()(i...) returns: a value for const storage, a proxy
     *  otherwise. */
    using access_type =
        std::conditional_t<is_const, arithmetic_type, reference_type>;

    static_assert(dimensionality >= 1, "an accessor has at least 1 dimension");

    friend class range<reduced_row_major>;

protected:
    static constexpr std::size_t num_strides = dimensionality - 1;

    /** size + storage + explicit row strides (outermost first). */
    constexpr reduced_row_major(length_type size, storage_type *storage,
                                storage_stride_type stride)
        : size_(detail::to_fixed(size)),
          storage_{storage},
          stride_(detail::to_fixed(stride))
    {}

    /** size + storage + strides given as individual integers. */
    template <typename... Strides,
              typename = std::enable_if_t<
                  (sizeof...(Strides) > 0) &&
                  sizeof...(Strides) == num_strides &&
                  detail::are_all_integral<Strides...>::value>>
    constexpr reduced_row_major(length_type size, storage_type *storage,
                                Strides... strides)
        : reduced_row_major{
              size, storage,
              storage_stride_type{{static_cast<size_type>(strides)...}}}
    {}

    /** size + storage: dense (gap-free) strides. */
    constexpr reduced_row_major(length_type size, storage_type *storage)
        : size_(detail::to_fixed(size)),
          storage_{storage},
          stride_(dense_stride(size))
    {}

    constexpr reduced_row_major()
        : reduced_row_major{length_type{}, nullptr, storage_stride_type{}}
    {}

public:
    /** Same view with read-only storage. */
    GKO_ACC_ATTRIBUTES constexpr range<const_accessor> to_const() const
    {
        return range<const_accessor>{detail::to_std(size_), storage_,
                                     detail::to_std(stride_)};
    }

    /** Element access. */
    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr std::enable_if_t<
        detail::are_all_integral<Indices...>::value &&
            sizeof...(Indices) == dimensionality,
        access_type>
    operator()(Indices... indices) const
    {
        return make_access(
            storage_ + linear_index<0>(static_cast<size_type>(indices)...),
            std::integral_constant<bool, is_const>{});
    }

    /** Sub-range: one index_span per dimension. Strides are kept. */
    template <typename... Spans>
    GKO_ACC_ATTRIBUTES constexpr std::enable_if_t<
        detail::are_all_spans<Spans...>::value &&
            sizeof...(Spans) == dimensionality,
        range<reduced_row_major>>
    operator()(Spans... spans) const
    {
        return range<reduced_row_major>{
            length_type{{spans.length()...}},
            storage_ + linear_index<0>(spans.begin...),
            detail::to_std(stride_)};
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type length(
        size_type dimension) const
    {
        return dimension < static_cast<size_type>(dimensionality)
                   ? size_[static_cast<std::size_t>(dimension)]
                   : 1;
    }

    /** Row strides, outermost first (dimensionality - 1 entries). */
    constexpr storage_stride_type get_stride() const
    {
        return detail::to_std(stride_);
    }

    /** Stride of dimension `d` (d < dimensionality - 1); device-usable. */
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type stride(
        std::size_t d) const
    {
        return stride_[d];
    }

    constexpr length_type get_size() const { return detail::to_std(size_); }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr storage_type *
    get_stored_data() const
    {
        return storage_;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr const storage_type *
    get_const_storage() const
    {
        return storage_;
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

    GKO_ACC_ATTRIBUTES static constexpr arithmetic_type make_access(
        storage_type *ptr, std::true_type)
    {
        return static_cast<arithmetic_type>(*ptr);
    }

    GKO_ACC_ATTRIBUTES static constexpr reference_type make_access(
        storage_type *ptr, std::false_type)
    {
        return reference_type{ptr};
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

    detail::fixed_array<size_type, dimensionality> size_;
    storage_type *storage_;
    detail::fixed_array<size_type, num_strides> stride_;
};


template <std::size_t D, typename A, typename S>
constexpr std::size_t reduced_row_major<D, A, S>::dimensionality;

template <std::size_t D, typename A, typename S>
constexpr bool reduced_row_major<D, A, S>::is_const;


}  // namespace acc
}  // namespace gko
