// gko::acc::scaled_reduced_row_major<Dim, Ar, St, ScalarMask>: like
// reduced_row_major, plus one arithmetic-type scalar per index combination of
// the dimensions selected by ScalarMask (bit d set = the scalar depends on
// dimension d): read = static_cast<Ar>(stored) * scalar, write =
// static_cast<St>(value / scalar). This lets integer or low-precision storage
// cover a wider range. Part of the accessor family the reference's README
// points to (`/root/reference/README.md:19`); SURVEY.md §8f N4.
#pragma once

#include "index_span.hpp"
#include "range.hpp"
#include "utils.hpp"


namespace gko {
namespace acc {
namespace reference_class {


/** Proxy of one scaled element: value = stored * scalar. */
template <typename ArithmeticType, typename StorageType>
class scaled_reduced_storage {
public:
    using arithmetic_type = std::remove_cv_t<ArithmeticType>;
    using storage_type = StorageType;

    scaled_reduced_storage() = delete;
    ~scaled_reduced_storage() = default;
    scaled_reduced_storage(const scaled_reduced_storage &) = default;
    scaled_reduced_storage(scaled_reduced_storage &&) = default;

    GKO_ACC_ATTRIBUTES constexpr scaled_reduced_storage(
        storage_type *ptr, arithmetic_type scalar) noexcept
        : ptr_{ptr}, scalar_{scalar}
    {}

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr operator arithmetic_type()
        const
    {
        return static_cast<arithmetic_type>(*ptr_) * scalar_;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type operator=(
        arithmetic_type value) const
    {
        *ptr_ = static_cast<storage_type>(value / scalar_);
        return value;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type operator=(
        const scaled_reduced_storage &other) const
    {
        return *this = static_cast<arithmetic_type>(other);
    }

#define GKO_ACC_SCALED_BINARY(_op)                                            \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const scaled_reduced_storage &a,                             \
                 const scaled_reduced_storage &b)                             \
    {                                                                         \
        return static_cast<arithmetic_type>(a)                                \
            _op static_cast<arithmetic_type>(b);                              \
    }                                                                         \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const scaled_reduced_storage &a, const arithmetic_type &b)   \
    {                                                                         \
        return static_cast<arithmetic_type>(a) _op b;                         \
    }                                                                         \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const arithmetic_type &a, const scaled_reduced_storage &b)   \
    {                                                                         \
        return a _op static_cast<arithmetic_type>(b);                         \
    }
    GKO_ACC_SCALED_BINARY(+)
    GKO_ACC_SCALED_BINARY(-)
    GKO_ACC_SCALED_BINARY(*)
    GKO_ACC_SCALED_BINARY(/)
#undef GKO_ACC_SCALED_BINARY

#define GKO_ACC_SCALED_COMPOUND(_op, _bin)                                    \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const scaled_reduced_storage &a, const arithmetic_type &b)   \
    {                                                                         \
        return a = static_cast<arithmetic_type>(a) _bin b;                    \
    }
    GKO_ACC_SCALED_COMPOUND(+=, +)
    GKO_ACC_SCALED_COMPOUND(-=, -)
    GKO_ACC_SCALED_COMPOUND(*=, *)
    GKO_ACC_SCALED_COMPOUND(/=, /)
#undef GKO_ACC_SCALED_COMPOUND

private:
    storage_type *ptr_;
    arithmetic_type scalar_;
};


}  // namespace reference_class


template <std::size_t Dimensionality, typename ArithmeticType,
          typename StorageType, std::uint64_t ScalarMask>
class scaled_reduced_row_major {
public:
    using arithmetic_type = std::remove_cv_t<ArithmeticType>;
    using storage_type = StorageType;
    static constexpr std::size_t dimensionality = Dimensionality;
    static constexpr std::uint64_t scalar_mask = ScalarMask;
    static constexpr bool is_const = std::is_const<storage_type>::value;
    using scalar_type =
        std::conditional_t<is_const, const arithmetic_type, arithmetic_type>;

    using const_accessor =
        scaled_reduced_row_major<dimensionality, arithmetic_type,
                                 const storage_type, ScalarMask>;
    using length_type = std::array<size_type, dimensionality>;
    using storage_stride_type = std::array<size_type, dimensionality - 1>;
    using reference_type =
        reference_class::scaled_reduced_storage<arithmetic_type, storage_type>;
    using access_type =
        std::conditional_t<is_const, arithmetic_type, reference_type>;

    static_assert(dimensionality >= 1, "an accessor has at least 1 dimension");
    static_assert(ScalarMask != 0 && ScalarMask < (std::uint64_t{1} << Dimensionality),
                  "the mask selects at least one existing dimension");

    friend class range<scaled_reduced_row_major>;

protected:
    /**
     * size, storage + its row strides, scalars. The scalar array is dense and
     * row-major over the masked dimensions (e.g. mask 0b10 on a 2-D range:
     * one scalar per column index... bit d = dimension d, dimension 0 is the
     * outermost index).
     */
    constexpr scaled_reduced_row_major(length_type size, storage_type *storage,
                                       storage_stride_type storage_stride,
                                       scalar_type *scalar)
        : size_(detail::to_fixed(size)),
          storage_{storage},
          stride_(detail::to_fixed(storage_stride)),
          scalar_{scalar}
    {}

    constexpr scaled_reduced_row_major(length_type size, storage_type *storage,
                                       scalar_type *scalar)
        : size_(detail::to_fixed(size)),
          storage_{storage},
          stride_(dense_stride(size)),
          scalar_{scalar}
    {}

public:
    GKO_ACC_ATTRIBUTES constexpr range<const_accessor> to_const() const
    {
        return range<const_accessor>{detail::to_std(size_), storage_,
                                     detail::to_std(stride_), scalar_};
    }

    /** Scalar belonging to the given element indices. */
    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type
    read_scalar_masked(Indices... indices) const
    {
        return scalar_[scalar_index<0>(size_type{0},
                                       static_cast<size_type>(indices)...)];
    }

    /** Overwrites the scalar belonging to the given element indices. */
    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr void write_scalar_masked(
        arithmetic_type value, Indices... indices) const
    {
        scalar_[scalar_index<0>(size_type{0},
                                static_cast<size_type>(indices)...)] = value;
    }

    template <typename... Indices>
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr std::enable_if_t<
        detail::are_all_integral<Indices...>::value &&
            sizeof...(Indices) == dimensionality,
        access_type>
    operator()(Indices... indices) const
    {
        return make_access(
            storage_ + linear_index<0>(static_cast<size_type>(indices)...),
            read_scalar_masked(indices...),
            std::integral_constant<bool, is_const>{});
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr size_type length(
        size_type dimension) const
    {
        return dimension < static_cast<size_type>(dimensionality)
                   ? size_[static_cast<std::size_t>(dimension)]
                   : 1;
    }

    constexpr storage_stride_type get_stride() const
    {
        return detail::to_std(stride_);
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

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr scalar_type *get_scalar() const
    {
        return scalar_;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr const arithmetic_type *
    get_const_scalar() const
    {
        return scalar_;
    }

private:
    static constexpr detail::fixed_array<size_type, dimensionality - 1>
    dense_stride(const length_type &size)
    {
        detail::fixed_array<size_type, dimensionality - 1> stride{};
        size_type running = 1;
        for (std::size_t d = dimensionality - 1; d > 0; --d) {
            running *= size[d];
            stride[d - 1] = running;
        }
        return stride;
    }

    GKO_ACC_ATTRIBUTES static constexpr arithmetic_type make_access(
        storage_type *ptr, arithmetic_type scalar, std::true_type)
    {
        return static_cast<arithmetic_type>(*ptr) * scalar;
    }

    GKO_ACC_ATTRIBUTES static constexpr reference_type make_access(
        storage_type *ptr, arithmetic_type scalar, std::false_type)
    {
        return reference_type{ptr, scalar};
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

    // dense row-major index over the masked dimensions only
    template <std::size_t Dim>
    GKO_ACC_ATTRIBUTES constexpr size_type scalar_index(size_type acc) const
    {
        static_assert(Dim == dimensionality, "wrong number of indices");
        return acc;
    }

    template <std::size_t Dim, typename... Rest>
    GKO_ACC_ATTRIBUTES constexpr size_type scalar_index(size_type acc,
                                                        size_type index,
                                                        Rest... rest) const
    {
        return scalar_index<Dim + 1>(
            (ScalarMask >> Dim) & 1 ? acc * size_[Dim] + index : acc, rest...);
    }

    detail::fixed_array<size_type, dimensionality> size_;
    storage_type *storage_;
    detail::fixed_array<size_type, dimensionality - 1> stride_;
    scalar_type *scalar_;
};


template <std::size_t D, typename A, typename S, std::uint64_t M>
constexpr std::size_t scaled_reduced_row_major<D, A, S, M>::dimensionality;

template <std::size_t D, typename A, typename S, std::uint64_t M>
constexpr bool scaled_reduced_row_major<D, A, S, M>::is_const;


}  // namespace acc
}  // namespace gko
