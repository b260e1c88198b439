// Proxy reference returned by non-const reduced-precision accessors.
//
// Contract (exercised by
// `/root/reference/cuda/gemv_kernels.cuh:86-87,108-110`,
// `cuda/trsv_kernels.cuh:641,651,672-673,683`):
//   * reading converts the stored value:  Ar v = ref  ==  static_cast<Ar>(*p)
//   * writing narrows:                    ref = v     ==  *p = static_cast<St>(v)
//   * + - * / with Ar values and with other proxies yield exactly Ar
//   * a proxy captured with `auto` stays a proxy and re-reads memory on use.
#pragma once

#include "utils.hpp"


namespace gko {
namespace acc {
namespace reference_class {


template <typename ArithmeticType, typename StorageType>
class reduced_storage {
public:
    using arithmetic_type = std::remove_cv_t<ArithmeticType>;
    using storage_type = StorageType;

    reduced_storage() = delete;
    ~reduced_storage() = default;
    reduced_storage(const reduced_storage &) = default;
    reduced_storage(reduced_storage &&) = default;

    GKO_ACC_ATTRIBUTES constexpr explicit reduced_storage(
        storage_type *ptr) noexcept
        : ptr_{ptr}
    {}

    /** Widening read. */
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr operator arithmetic_type()
        const
    {
        return static_cast<arithmetic_type>(*ptr_);
    }

    /** Narrowing write. Returns the unrounded value like a built-in `=`
     *  chained through the arithmetic type would. */
    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type operator=(
        arithmetic_type value) const
    {
        *ptr_ = static_cast<storage_type>(value);
        return value;
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type operator=(
        const reduced_storage &other) const
    {
        return *this = static_cast<arithmetic_type>(other);
    }

    GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type operator=(
        reduced_storage &&other) const
    {
        return *this = static_cast<arithmetic_type>(other);
    }

    GKO_ACC_ATTRIBUTES constexpr storage_type *address() const noexcept
    {
        return ptr_;
    }

#define GKO_ACC_PROXY_BINARY(_op)                                             \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const reduced_storage &a, const reduced_storage &b)          \
    {                                                                         \
        return static_cast<arithmetic_type>(a)                                \
            _op static_cast<arithmetic_type>(b);                              \
    }                                                                         \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const reduced_storage &a, const arithmetic_type &b)          \
    {                                                                         \
        return static_cast<arithmetic_type>(a) _op b;                         \
    }                                                                         \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const arithmetic_type &a, const reduced_storage &b)          \
    {                                                                         \
        return a _op static_cast<arithmetic_type>(b);                         \
    }
    GKO_ACC_PROXY_BINARY(+)
    GKO_ACC_PROXY_BINARY(-)
    GKO_ACC_PROXY_BINARY(*)
    GKO_ACC_PROXY_BINARY(/)
#undef GKO_ACC_PROXY_BINARY

#define GKO_ACC_PROXY_COMPOUND(_op, _bin)                                     \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const reduced_storage &a, const reduced_storage &b)          \
    {                                                                         \
        return a = static_cast<arithmetic_type>(a)                            \
                   _bin static_cast<arithmetic_type>(b);                      \
    }                                                                         \
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type        \
    operator _op(const reduced_storage &a, const arithmetic_type &b)          \
    {                                                                         \
        return a = static_cast<arithmetic_type>(a) _bin b;                    \
    }
    GKO_ACC_PROXY_COMPOUND(+=, +)
    GKO_ACC_PROXY_COMPOUND(-=, -)
    GKO_ACC_PROXY_COMPOUND(*=, *)
    GKO_ACC_PROXY_COMPOUND(/=, /)
#undef GKO_ACC_PROXY_COMPOUND

    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type
    operator+(const reduced_storage &a)
    {
        return +static_cast<arithmetic_type>(a);
    }
    friend GKO_ACC_ATTRIBUTES GKO_ACC_INLINE constexpr arithmetic_type
    operator-(const reduced_storage &a)
    {
        return -static_cast<arithmetic_type>(a);
    }

private:
    storage_type *ptr_;
};


}  // namespace reference_class
}  // namespace acc
}  // namespace gko
