// C-ABI GEMV entry points (include/accblas.h). Each builds the views the way
// the reference launchers do (`/root/reference/cuda/gemv_kernels.cuh:136-147`
// plain, `:168-193` accessor) and hands them to the gfx950 kernels.
#include <array>

#include "accblas_internal.hpp"
#include "gemv_kernels.hpp"


namespace accblas {
inline dev::gemv_skew_mode gemv_skew_option()
{
    return static_cast<dev::gemv_skew_mode>(gemv_skew_option_value());
}
}  // namespace accblas

extern "C" {


int accblas_gemv(accblas_dtype vt, int64_t rows, int64_t cols, double alpha,
                 const void *A, int64_t lda, const void *x, int64_t incx,
                 double beta, void *res, int64_t incres, void *stream)
{
    const int status = accblas::validate_gemv(rows, cols, A, lda, x, incx, res, incres);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(vt, [&](auto tag) {
        using value_type = typename decltype(tag)::type;
        using namespace accblas::dev;
        const plain_view<const value_type> m_view{
            static_cast<const value_type *>(A), {rows, cols}, lda};
        const plain_view<const value_type> x_view{
            static_cast<const value_type *>(x), {cols, 1}, incx};
        const plain_view<value_type> res_view{static_cast<value_type *>(res),
                                              {rows, 1}, incres};
        return accblas::check_hip(
            launch_gemv(static_cast<value_type>(alpha), m_view, x_view,
                        static_cast<value_type>(beta), res_view,
                        accblas::as_stream(stream), accblas::gemv_skew_option(),
                        accblas::gemv_vector_rows_option_value() != 0),
            "gemv launch");
    });
}


int accblas_acc_gemv(accblas_dtype ar, accblas_dtype st, int64_t rows,
                     int64_t cols, double alpha, const void *A, int64_t lda,
                     const void *x, int64_t incx, double beta, void *res,
                     int64_t incres, void *stream)
{
    const int status = accblas::validate_gemv(rows, cols, A, lda, x, incx, res, incres);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(ar, [&](auto ar_tag) {
        using ar_type = typename decltype(ar_tag)::type;
        return accblas::dispatch_type(st, [&](auto st_tag) {
            using st_type = typename decltype(st_tag)::type;
            // Accessor setup, as cuda/gemv_kernels.cuh:178-189
            constexpr std::size_t dimensionality{2};
            using accessor =
                gko::acc::reduced_row_major<dimensionality, ar_type, st_type>;
            using range = gko::acc::range<accessor>;
            using c_range =
                gko::acc::range<typename accessor::const_accessor>;
            using size_array = std::array<gko::acc::size_type, 2>;
            using stride_array = std::array<gko::acc::size_type, 1>;
            const auto m_acc =
                c_range(size_array{{rows, cols}},
                        static_cast<const st_type *>(A), stride_array{{lda}});
            const auto x_acc =
                c_range(size_array{{cols, 1}}, static_cast<const st_type *>(x),
                        stride_array{{incx}});
            const auto res_acc =
                range(size_array{{rows, 1}}, static_cast<st_type *>(res),
                      stride_array{{incres}});
            return accblas::check_hip(
                accblas::dev::launch_gemv(static_cast<ar_type>(alpha), m_acc,
                                          x_acc, static_cast<ar_type>(beta),
                                          res_acc, accblas::as_stream(stream),
                                          accblas::gemv_skew_option(),
                                          accblas::gemv_vector_rows_option_value() != 0),
                "acc_gemv launch");
        });
    });
}


}  // extern "C"
