// Shared pieces of the TRSV C-ABI translation units. The 64 kernel variants
// (uplo x diag x vector/scalar x 8 type pairs) are split over three TUs
// (accblas_trsv.hip: plain + entry points, accblas_trsv_f32.hip /
// accblas_trsv_f64.hip: accessor kernels by arithmetic type) so that they
// compile in parallel.
#pragma once

#include <array>
#include <cstdint>

#include "accblas_internal.hpp"
#include "trsv_kernels.hpp"


namespace accblas {


/** Hand-off workspace (8-byte granules) owned by the library per (device,
 *  stream); defined in accblas_trsv.hip. */
int trsv_get_workspace(hipStream_t stream, std::size_t bytes,
                       dev::granule_word **out);

/** acc_trsv for one arithmetic type, all storage types. */
int acc_trsv_f32(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream);
int acc_trsv_f64(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream);


template <typename ar_type>
int acc_trsv_typed(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                   int64_t n, const void *A, int64_t lda, void *x,
                   int64_t incx, uint32_t *helper, void *stream)
{
    return dispatch_type(st, [&](auto st_tag) {
        using st_type = typename decltype(st_tag)::type;
        // Accessor setup, as cuda/trsv_kernels.cuh:923-933
        constexpr std::size_t dimensionality{2};
        using accessor =
            gko::acc::reduced_row_major<dimensionality, ar_type, st_type>;
        using range = gko::acc::range<accessor>;
        using c_range = gko::acc::range<typename accessor::const_accessor>;
        using size_array = std::array<gko::acc::size_type, 2>;
        using stride_array = std::array<gko::acc::size_type, 1>;
        const auto m_acc =
            c_range(size_array{{n, n}}, static_cast<const st_type *>(A),
                    stride_array{{lda}});
        const auto x_acc = range(size_array{{n, 1}}, static_cast<st_type *>(x),
                                 stride_array{{incx}});
        dev::granule_word *granules = nullptr;
        const int ws_status = trsv_get_workspace(
            as_stream(stream), dev::trsv_workspace_bytes<st_type>(n),
            &granules);
        if (ws_status != ACCBLAS_OK) {
            return ws_status;
        }
        return check_hip(
            dev::launch_trsv<ar_type>(uplo == ACCBLAS_UPPER,
                                      diag == ACCBLAS_UNIT, m_acc, x_acc,
                                      helper, granules, as_stream(stream)),
            "acc_trsv launch");
    });
}


}  // namespace accblas
