// Shared pieces of the TRSV C-ABI translation units. The kernel variants
// (uplo x diag x {64-row vector / scalar / relay, 128-row} x 8 type pairs) are
// split over three TUs
// (accblas_trsv.hip: plain + entry points, accblas_trsv_f32.hip /
// accblas_trsv_f64.hip: accessor kernels by arithmetic type) so that they
// compile in parallel.
#pragma once

#include <array>
#include <cstdint>

#include <mutex>

#include "accblas_internal.hpp"
#include "trsv_kernels.hpp"
#include "trsv_small_kernels.hpp"
#include "trsv_wide_kernels.hpp"


namespace accblas {


/** Bytes in front of the granules of a TRSV workspace (holds `trsv_state`). */
constexpr std::size_t trsv_workspace_header_bytes = 4096;

/** Total bytes of a workspace for n unknowns of storage type St. */
template <typename St>
constexpr std::size_t trsv_workspace_total_bytes(int64_t n)
{
    return trsv_workspace_header_bytes + dev::trsv_workspace_bytes<St>(n);
}

/**
 * The library-owned hand-off workspace of a (device, stream), leased for the
 * duration of one launch: the lock is held until the lease is dropped, so
 * another host thread cannot grow (and retire) the buffer between "pointer
 * taken" and "kernel enqueued". Buffers that were ever handed to a launch are
 * never freed behind the caller's back: growth allocates a new one and
 * retires the old one until `accblas_trsv_release_workspace` (defined in
 * accblas_trsv.hip).
 */
class trsv_workspace_lease {
public:
    trsv_workspace_lease() = default;
    trsv_workspace_lease(const trsv_workspace_lease &) = delete;
    trsv_workspace_lease &operator=(const trsv_workspace_lease &) = delete;

    /** Status of include/accblas.h; on success `session()` is valid. */
    int acquire(hipStream_t stream, std::size_t granule_bytes);
    const dev::trsv_session &session() const { return session_; }

private:
    std::unique_lock<std::mutex> lock_;
    dev::trsv_session session_{};
};

/** Compute units of the current device (cached; the 128-row kernel splits the
 *  row blocks that cannot be resident from the start). */
int trsv_num_cus();

/** Option "trsv_wide": 0 = 64-row kernel only, 1 = 128-row kernel by the
 *  measured rule (default), 2 = 128-row kernel whenever the operands allow. */
int trsv_wide_option_value();
/** Option "trsv_wide_split_from": 0 = the first split row block by the rule
 *  (`trsv_wide_split_for`, default), k > 0 = block k (tests, experiments). */
int trsv_split_from_option_value();
/** Option "trsv_prefetch": 0 = no prefetching workgroups, 1 = by the measured
 *  rule (default, `trsv_wide_prefetchers_for`), 2 = whenever CUs are idle. */
int trsv_prefetch_option_value();
/** Option "trsv_wide_persistent": 0 = the 128-row kernel is launched with one
 *  workgroup per ticket, 1 = with one workgroup per CU drawing tickets until
 *  none is left when the solve has more tickets than CUs. */
int trsv_wide_persistent_option_value();
/** Option "trsv_relay": 0 never, 1 by the measured rule (default), 2 always. */
int trsv_relay_option_value();
/** Option "trsv_vector_rows": 0 = 16-byte loads of the matrix only for
 *  16-byte aligned bases and row strides (the rule until round 3), 1 = for
 *  every operand (default). */
int trsv_vector_rows_option_value();
/** Option "trsv_small": 0 = the blocked kernels at every size, 1 = systems of
 *  at most 128 unknowns by one workgroup without a workspace (default). */
int trsv_small_option_value();

/** Where the 128-row kernel is the faster one. 2- and 4-byte storage: from
 *  n = 3328 (26 row blocks) on. Rounds 2-4 switched at 4096
 *  (profiles/r02_trsv_ab_final.txt: 4-7 % faster there, 17-20 % at 8192, 27-50 %
 *  at 24000-65536), which left the reference's sweep 3-7 % slower at n = 3800 ...
 *  4000 than at 4100; measured in round 5 with the two kernels alternated solve
 *  by solve inside a 24000-wide matrix (profiles/r05_trsv_kernel_switch.txt):
 *  equal at 3200, the 128-row kernel 1-4 % faster at 3400 ... 3600 and 3-7 % at
 *  3800 ... 4000 (fp64 and fp32 arithmetic; 2-byte storage from 3000 on). 8-byte
 *  storage: from n = 61440 (`ACCBLAS_TRSV_WIDE_MIN_N_8BYTE`). Until round
 *  3 the 128-row kernel (two half-width steps in flight are all its registers
 *  allow) won below 17000 and beyond 32768; since round 4 the 64-row kernel's
 *  relay variant keeps its ring of panels in flight and its workgroups draw
 *  tickets until none is left (see `kernel::trsv`), and it is the faster one at
 *  every size measured (profiles/r04_trsv_relay_ring.txt, us, 64-row relay |
 *  128-row with prefetchers or split blocks): 4096 82 | 89, 8192 160 | 168,
 *  12288 244 | 267, 16384 329 | 392, 24000 523 | 677, 32768 796 | 978, 40000
 *  1139 | 1270, 49152 1684 | 1726, 57344 2074 | 2082, 61440 2359 | 2333, 65536
 *  2785 | 2705 (within 3 % of each other from 52000 on). */
#ifndef ACCBLAS_TRSV_WIDE_MIN_N
#define ACCBLAS_TRSV_WIDE_MIN_N 3328
#endif
/** 8-byte storage: the 64-row relay kernel up to here (20-30 % ahead at 16384 ...
 *  40000), the 128-row kernel with its split row blocks beyond - 2-3 % faster
 *  from 61440 on in rounds 4 and 5 (profiles/r05_trsv_f64_storage_unfolded.txt:
 *  65536 lower 2595 against 2666 us, upper 2616 against 2655). */
#ifndef ACCBLAS_TRSV_WIDE_MIN_N_8BYTE
#define ACCBLAS_TRSV_WIDE_MIN_N_8BYTE 61440
#endif


/**
 * Picks the kernel and launches: `kernel::trsv_wide` (128-row blocks) for
 * large aligned systems, `kernel::trsv` otherwise. `external` = a
 * caller-owned, initialised workspace (accblas_*_trsv_ws) or NULL for the
 * library-owned one of the stream.
 */
template <typename ar_type, typename st_type, typename MtxView, typename XView>
int trsv_dispatch(bool upper, bool unit_diag, int64_t n, const MtxView &mtx,
                  const XView &x, uint32_t *helper, void *stream,
                  const dev::trsv_session *external = nullptr)
{
    if (n <= 0) {
        return ACCBLAS_OK;
    }
    const std::size_t bytes = dev::trsv_workspace_bytes<st_type>(n);
    if (external != nullptr &&
        static_cast<std::size_t>(external->capacity_words) *
                sizeof(dev::granule_word) <
            bytes) {
        return set_error(ACCBLAS_ERR_INVALID_ARGUMENT,
                         "trsv: workspace smaller than "
                         "accblas_trsv_workspace_bytes(st, n)");
    }
    if (n <= dev::trsv_small_max_n && trsv_small_option_value() != 0) {
        // one workgroup, nothing handed over through memory: no workspace is
        // leased (or allocated) for such a solve, the caller's stays untouched
        return check_hip(dev::launch_trsv_small<ar_type>(upper, unit_diag, mtx,
                                                         x, helper,
                                                         as_stream(stream)),
                         "trsv (one workgroup) launch");
    }
    trsv_workspace_lease lease;
    dev::trsv_session session{};
    if (external != nullptr) {
        session = *external;
    } else {
        const int status = lease.acquire(as_stream(stream), bytes);
        if (status != ACCBLAS_OK) {
            return status;
        }
        session = lease.session();
    }
    // the key the launch's workgroups recognise each other by (`trsv_state`)
    session.serial = dev::trsv_next_serial();
    const int wide = trsv_wide_option_value();
    const bool any_alignment = trsv_vector_rows_option_value() != 0;
    const bool use_wide =
        wide != 0 && dev::trsv_wide_applicable(upper, mtx, any_alignment) &&
        (wide == 2 ||
         (sizeof(st_type) < 8 ? n >= ACCBLAS_TRSV_WIDE_MIN_N
                              : n >= ACCBLAS_TRSV_WIDE_MIN_N_8BYTE));
    if (use_wide) {
        return check_hip(
            dev::launch_trsv_wide<ar_type>(upper, unit_diag, mtx, x, helper,
                                           session, as_stream(stream),
                                           trsv_num_cus(),
                                           trsv_split_from_option_value(),
                                           trsv_prefetch_option_value(),
                                           trsv_wide_persistent_option_value()),
            "trsv (128-row) launch");
    }
    return check_hip(dev::launch_trsv<ar_type>(upper, unit_diag, mtx, x, helper,
                                               session, as_stream(stream),
                                               trsv_relay_option_value(),
                                               trsv_num_cus(), any_alignment),
                     "trsv launch");
}

/** acc_trsv for one arithmetic type, all storage types. */
int acc_trsv_f32(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream,
                 const dev::trsv_session *external);
int acc_trsv_f64(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                 int64_t n, const void *A, int64_t lda, void *x, int64_t incx,
                 uint32_t *helper, void *stream,
                 const dev::trsv_session *external);


template <typename ar_type>
int acc_trsv_typed(accblas_dtype st, accblas_uplo uplo, accblas_diag diag,
                   int64_t n, const void *A, int64_t lda, void *x,
                   int64_t incx, uint32_t *helper, void *stream,
                   const dev::trsv_session *external)
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
                                               x_acc, helper, stream, external);
    });
}


}  // namespace accblas
