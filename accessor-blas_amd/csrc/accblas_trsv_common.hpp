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
#include "trsv_split_kernels.hpp"
#include "trsv_wide_kernels.hpp"


namespace accblas {


/** Hand-off workspace (8-byte granules) owned by the library per (device,
 *  stream); defined in accblas_trsv.hip. */
int trsv_get_workspace(hipStream_t stream, std::size_t bytes,
                       dev::granule_word **out);

/** The same workspace as `kernel::trsv` sees it (state header + granules). */
int trsv_get_session(hipStream_t stream, std::size_t bytes,
                     dev::trsv_session *out);

/** Compute units of the current device (cached). */
int trsv_num_cus();

/** Options "trsv_split" / "trsv_wide" (accblas_set_option). */
bool trsv_split_enabled();
bool trsv_wide_enabled();
/** Option "trsv_relay": 0 never, 1 by the measured rule (default), 2 always. */
int trsv_relay_option_value();


/** Picks the role-split kernel for large aligned systems, `kernel::trsv`
 *  otherwise, and launches. */
template <typename ar_type, typename st_type, typename MtxView, typename XView>
int trsv_dispatch(bool upper, bool unit_diag, int64_t n, const MtxView &mtx,
                  const XView &x, uint32_t *helper, void *stream)
{
    if (n <= 0) {
        return ACCBLAS_OK;
    }
    const std::size_t bytes =
        std::max(dev::trsv_workspace_bytes<st_type>(n),
                 dev::trsv_split_workspace_bytes<ar_type, st_type>(n));
    // the opt-in kernels keep the two-launch protocol (trsv_init zeroes the
    // granules before every solve)
    if (trsv_wide_enabled() || trsv_split_enabled()) {
        dev::granule_word *granules = nullptr;
        const int ws_status =
            trsv_get_workspace(as_stream(stream), bytes, &granules);
        if (ws_status != ACCBLAS_OK) {
            return ws_status;
        }
        bool launched = false;
        if (trsv_wide_enabled()) {
            const int wide_status = check_hip(
                dev::try_launch_trsv_wide<ar_type>(upper, unit_diag, mtx, x,
                                                   helper, granules,
                                                   as_stream(stream), &launched),
                "trsv (128-row) launch");
            if (wide_status != ACCBLAS_OK || launched) {
                return wide_status;
            }
        }
        if (trsv_split_enabled()) {
            const int split_status = check_hip(
                dev::try_launch_trsv_split<ar_type>(
                    upper, unit_diag, mtx, x, helper, granules, trsv_num_cus(),
                    as_stream(stream), &launched),
                "trsv (split) launch");
            if (split_status != ACCBLAS_OK || launched) {
                return split_status;
            }
        }
    }
    dev::trsv_session session{};
    const int status = trsv_get_session(as_stream(stream), bytes, &session);
    if (status != ACCBLAS_OK) {
        return status;
    }
    return check_hip(dev::launch_trsv<ar_type>(upper, unit_diag, mtx, x, helper,
                                               session, as_stream(stream),
                                               trsv_relay_option_value()),
                     "trsv launch");
}

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
        return trsv_dispatch<ar_type, st_type>(uplo == ACCBLAS_UPPER,
                                               diag == ACCBLAS_UNIT, n, m_acc,
                                               x_acc, helper, stream);
    });
}


}  // namespace accblas
