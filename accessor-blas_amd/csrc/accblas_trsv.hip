// C-ABI TRSV entry points (include/accblas.h). Views are built as the
// reference launchers do (`/root/reference/cuda/trsv_kernels.cuh:455-488`
// plain, `:918-961` accessor).
#include <array>

#include "accblas_internal.hpp"
#include "trsv_kernels.hpp"


namespace {


int validate_trsv(accblas_uplo uplo, accblas_diag diag, int64_t n,
                  const void *A, int64_t lda, const void *x, int64_t incx,
                  const uint32_t *helper)
{
    if (uplo != ACCBLAS_UPPER && uplo != ACCBLAS_LOWER) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: unknown uplo");
    }
    if (diag != ACCBLAS_UNIT && diag != ACCBLAS_NON_UNIT) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: unknown diag");
    }
    if (n < 0) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT, "trsv: n < 0");
    }
    if (lda < n) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: row stride smaller than n");
    }
    if (incx < 1) {
        return accblas::set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                                  "trsv: increment must be >= 1");
    }
    if (helper == nullptr) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv: helper is NULL");
    }
    if (n > 0 && (A == nullptr || x == nullptr)) {
        return accblas::set_error(ACCBLAS_ERR_NULL_POINTER,
                                  "trsv: A or x is NULL");
    }
    return ACCBLAS_OK;
}


}  // namespace


extern "C" {


int accblas_trsv(accblas_dtype vt, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream)
{
    const int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(vt, [&](auto tag) {
        using value_type = typename decltype(tag)::type;
        using namespace accblas::dev;
        const plain_view<const value_type> m_view{
            static_cast<const value_type *>(A), {n, n}, lda};
        const plain_view<value_type> x_view{static_cast<value_type *>(x),
                                            {n, 1}, incx};
        return accblas::check_hip(
            launch_trsv<value_type>(uplo == ACCBLAS_UPPER, diag == ACCBLAS_UNIT,
                                    m_view, x_view, helper,
                                    accblas::as_stream(stream)),
            "trsv launch");
    });
}


int accblas_acc_trsv(accblas_dtype ar, accblas_dtype st, accblas_uplo uplo,
                     accblas_diag diag, int64_t n, const void *A, int64_t lda,
                     void *x, int64_t incx, uint32_t *helper, void *stream)
{
    const int status = validate_trsv(uplo, diag, n, A, lda, x, incx, helper);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return accblas::dispatch_arithmetic(ar, [&](auto ar_tag) {
        using ar_type = typename decltype(ar_tag)::type;
        return accblas::dispatch_type(st, [&](auto st_tag) {
            using st_type = typename decltype(st_tag)::type;
            // Accessor setup, as cuda/trsv_kernels.cuh:923-933
            constexpr std::size_t dimensionality{2};
            using accessor =
                gko::acc::reduced_row_major<dimensionality, ar_type, st_type>;
            using range = gko::acc::range<accessor>;
            using c_range =
                gko::acc::range<typename accessor::const_accessor>;
            using size_array = std::array<gko::acc::size_type, 2>;
            using stride_array = std::array<gko::acc::size_type, 1>;
            const auto m_acc =
                c_range(size_array{{n, n}}, static_cast<const st_type *>(A),
                        stride_array{{lda}});
            const auto x_acc = range(size_array{{n, 1}},
                                     static_cast<st_type *>(x),
                                     stride_array{{incx}});
            return accblas::check_hip(
                accblas::dev::launch_trsv<ar_type>(
                    uplo == ACCBLAS_UPPER, diag == ACCBLAS_UNIT, m_acc, x_acc,
                    helper, accblas::as_stream(stream)),
                "acc_trsv launch");
        });
    });
}


}  // extern "C"
