// C-ABI entry points for `gko::acc::scaled_reduced_row_major` operands
// (include/accblas.h, "other accessor kinds"). The reference's README names the
// accessor family its kernels accept ("all others work accordingly",
// `/root/reference/README.md:19`); its own launchers only ever build
// `reduced_row_major` (`/root/reference/cuda/gemv_kernels.cuh:178-189`,
// `dot_kernels.cuh:233-243`). The kernels are the library's GEMV / dot
// templates (gemv_kernels.hpp, dot_kernels.hpp), which take any view: a scaled
// operand is read through its `operator()` - value = storage x scalar - on the
// element-wise paths, the other operands keep their 16-byte loads.
#include <array>
#include <cstdint>

#include <accessor/scaled_reduced_row_major.hpp>

#include "accblas_internal.hpp"
#include "dot_kernels.hpp"
#include "gemv_kernels.hpp"


namespace {


/** Storage types a scaled operand may have: integers and the two narrow
 *  floating-point types (a scale in front of fp64 storage buys nothing). */
template <typename F>
int dispatch_scaled_storage(accblas_dtype t, F &&f)
{
    switch (t) {
    case ACCBLAS_I16:
        return f(accblas::type_tag<std::int16_t>{});
    case ACCBLAS_I32:
        return f(accblas::type_tag<std::int32_t>{});
    case ACCBLAS_F16:
        return f(accblas::type_tag<accblas::f16>{});
    case ACCBLAS_F32:
        return f(accblas::type_tag<float>{});
    default:
        return accblas::set_error(
            ACCBLAS_ERR_UNSUPPORTED_TYPES,
            "scaled storage must be i16, i32, f16 or f32");
    }
}

/** Storage of the unscaled vectors beside it: fp32 or fp64. */
template <typename F>
int dispatch_vector_storage(accblas_dtype t, F &&f)
{
    switch (t) {
    case ACCBLAS_F32:
        return f(accblas::type_tag<float>{});
    case ACCBLAS_F64:
        return f(accblas::type_tag<double>{});
    default:
        return accblas::set_error(
            ACCBLAS_ERR_UNSUPPORTED_TYPES,
            "vectors beside a scaled operand are stored as f32 or f64");
    }
}


using size_array = std::array<gko::acc::size_type, 2>;
using stride_array = std::array<gko::acc::size_type, 1>;
// one scalar per column index (bit 1 of the mask = dimension 1)
constexpr std::uint64_t per_column = 0b10;


}  // namespace


extern "C" {


int accblas_scaled_acc_gemv(accblas_dtype ar, accblas_dtype a_st,
                            accblas_dtype v_st, int64_t rows, int64_t cols,
                            double alpha, const void *A, int64_t lda,
                            const void *scalars, const void *x, int64_t incx,
                            double beta, void *res, int64_t incres,
                            void *stream)
{
    const int status =
        accblas::validate_gemv(rows, cols, A, lda, x, incx, res, incres);
    if (status != ACCBLAS_OK) {
        return status;
    }
    if (rows > 0 && cols > 0 && scalars == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "scaled gemv: scalars is NULL");
    }
    return accblas::dispatch_arithmetic(ar, [&](auto ar_tag) {
        using ar_type = typename decltype(ar_tag)::type;
        return dispatch_scaled_storage(a_st, [&](auto a_tag) {
            using a_type = typename decltype(a_tag)::type;
            return dispatch_vector_storage(v_st, [&](auto v_tag) {
                using v_type = typename decltype(v_tag)::type;
                using mtx_accessor =
                    gko::acc::scaled_reduced_row_major<2, ar_type, a_type,
                                                       per_column>;
                using c_mtx =
                    gko::acc::range<typename mtx_accessor::const_accessor>;
                using vec_accessor =
                    gko::acc::reduced_row_major<2, ar_type, v_type>;
                using vec = gko::acc::range<vec_accessor>;
                using c_vec =
                    gko::acc::range<typename vec_accessor::const_accessor>;
                const auto m_acc =
                    c_mtx(size_array{{rows, cols}},
                          static_cast<const a_type *>(A), stride_array{{lda}},
                          static_cast<const ar_type *>(scalars));
                const auto x_acc =
                    c_vec(size_array{{cols, 1}},
                          static_cast<const v_type *>(x),
                          stride_array{{incx}});
                const auto res_acc =
                    vec(size_array{{rows, 1}}, static_cast<v_type *>(res),
                        stride_array{{incres}});
                return accblas::check_hip(
                    accblas::dev::launch_gemv(static_cast<ar_type>(alpha),
                                              m_acc, x_acc,
                                              static_cast<ar_type>(beta),
                                              res_acc,
                                              accblas::as_stream(stream)),
                    "scaled acc_gemv launch");
            });
        });
    });
}


int accblas_scaled_acc_dot(accblas_handle *handle, accblas_dtype ar,
                           accblas_dtype x_st, accblas_dtype y_st, int64_t n,
                           const void *x, int64_t incx, const void *x_scalar,
                           const void *y, int64_t incy, void *res,
                           void *stream)
{
    const int status =
        accblas::validate_dot(handle, n, x, incx, y, incy, res);
    if (status != ACCBLAS_OK) {
        return status;
    }
    if (n > 0 && x_scalar == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "scaled dot: x_scalar is NULL");
    }
    return accblas::dispatch_arithmetic(ar, [&](auto ar_tag) {
        using ar_type = typename decltype(ar_tag)::type;
        return dispatch_scaled_storage(x_st, [&](auto x_tag) {
            using x_type = typename decltype(x_tag)::type;
            return dispatch_vector_storage(y_st, [&](auto y_tag) {
                using y_type = typename decltype(y_tag)::type;
                using x_accessor =
                    gko::acc::scaled_reduced_row_major<2, ar_type, x_type,
                                                       per_column>;
                using c_x =
                    gko::acc::range<typename x_accessor::const_accessor>;
                using y_accessor =
                    gko::acc::reduced_row_major<2, ar_type, y_type>;
                using c_y =
                    gko::acc::range<typename y_accessor::const_accessor>;
                // (an (n, 1) range: its one column has one scalar)
                const auto x_acc =
                    c_x(size_array{{n, 1}}, static_cast<const x_type *>(x),
                        stride_array{{incx}},
                        static_cast<const ar_type *>(x_scalar));
                const auto y_acc =
                    c_y(size_array{{n, 1}}, static_cast<const y_type *>(y),
                        stride_array{{incy}});
                return accblas::check_hip(
                    accblas::dev::launch_dot<ar_type>(
                        handle->dot_ws,
                        handle->dot_mode == ACCBLAS_DOT_ATOMIC, x_acc, y_acc,
                        static_cast<ar_type *>(res),
                        accblas::as_stream(stream)),
                    "scaled acc_dot launch");
            });
        });
    });
}


}  // extern "C"
