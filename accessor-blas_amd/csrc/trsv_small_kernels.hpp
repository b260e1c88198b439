// TRSV for systems of at most 128 unknowns: ONE workgroup, no workspace.
//
// The reference's benchmark starts its sweep at n = 100 and a solver's coarse
// levels are of this size (`/root/reference/cuda/trsv_benchmark.cu:200-203`;
// kernels `cuda/trsv_kernels.cuh:69-235,263-432`). Such a solve is a single row
// block (or two) of the blocked kernels, and what they spend there is
// machinery that only a chain of blocks needs: a ticket and an epoch drawn with
// atomics (two dependent round trips before the first load), the diagonal
// tile INVERTED (3.3 us, so that later a block's turn is one mat-vec), hand-off
// words written through to memory, a "finished" count and the re-arming of the
// workspace before the kernel may end (two more round trips). With one
// workgroup nobody waits for anybody:
//   * all four waves stage the triangle in LDS (one batch of loads, in the
//     storage type, element-wise: any alignment, any row stride), the
//     right-hand side is loaded beside it;
//   * wave 0 solves by forward substitution IN THE WAVE: lane i owns row i and
//     its running right-hand side, the solved x_j goes from lane j to all lanes
//     through SGPRs (`v_readlane`), one fused multiply-add per step and lane; the
//     row's coefficients come from LDS a quarter of a tile ahead. ~35 cycles per
//     unknown, against 16 serial steps + two levels of matrix-core products +
//     four barriers for an inverse that would be used once;
//   * unknowns 64 .. 127: the 64 x 64 block below the first tile is applied to
//     x_0..63 by the same wave (lane = row, x broadcast from LDS), then the
//     second tile is substituted like the first.
// Upper systems are the lower solve of the index-reversed system (as in the
// blocked kernels), a non-unit diagonal multiplies by the reciprocal of the
// diagonal entry (formed by a division, off the dependency chain).
#pragma once

#include <hip/hip_runtime.h>

#include "trsv_kernels.hpp"


namespace accblas {
namespace dev {


constexpr index_type trsv_small_max_n = 128;


namespace kernel {


/**
 * x <- T^{-1} x for n <= 128, one workgroup of 256 threads, grid = 1.
 * Meaning as `kernel::trsv` (`cuda/trsv_kernels.cuh:527-692` lower / `:725-893`
 * upper): arithmetic in ArType, x stored - and, between the two 64-row halves,
 * re-read - as the storage type.
 */
template <bool Upper, bool UnitDiag, typename ArType, typename MtxView,
          typename XView>
__global__ __launch_bounds__(256) void trsv_small(MtxView mtx, XView x,
                                                  std::uint32_t *helper)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    constexpr int B = trsv_tile;  // 64
    constexpr int ld = B + 1;
    constexpr int Waves = 4;
    // tiles[0] = T00, tiles[1] = T11, tiles[2] = T10, all column-major
    __shared__ ar_type tiles[3][B * ld];
    __shared__ ar_type x_first[B];  // x_0..63 as stored, widened

    const int lane = lane_id();
    const int wave = wave_id();
    const index_type n = mtx.length(0);
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };
    const auto inside = [n](index_type logical) {
        return logical < n ? logical : n - 1;
    };
    const st_type *a_data = mtx_traits::data(mtx);
    const index_type lda = mtx_traits::stride(mtx);
    st_type *x_data = x_traits::data(x);
    const index_type x_inc = x_traits::stride(x);
    const bool two_tiles = n > B;

    // ---- right-hand sides and diagonal entries (wave 0, lane = row) ---------
    st_type rhs_raw[2] = {};
    st_type diag_raw[2] = {};
    if (wave == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const index_type row = inside(h * B + lane);
            rhs_raw[h] = x_data[phys(row) * x_inc];
            if (!UnitDiag) {
                diag_raw[h] = a_data[phys(row) * lda + phys(row)];
            }
        }
    }
    // ---- stage the triangle: a wave reads one tile row per load (lane =
    // column), every load unconditional (clamped into the matrix), all of a
    // wave's loads in flight together, conversion behind them ------------------
    {
        constexpr int RS = B / Waves;  // tile rows a wave stages per tile
        st_type raw00[RS], raw11[RS], raw10[RS];
        const index_type col0 = phys(inside(lane));
        const index_type col1 = phys(inside(B + lane));
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int r = wave + i * Waves;
            const st_type *row0 = a_data + phys(inside(r)) * lda;
            raw00[i] = row0[col0];
        }
        if (two_tiles) {
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                const int r = wave + i * Waves;
                const st_type *row1 = a_data + phys(inside(B + r)) * lda;
                raw11[i] = row1[col1];
                raw10[i] = row1[col0];
            }
        }
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int r = wave + i * Waves;
            // (entries of rows or columns that do not exist: zero, the row
            // then passes its right-hand side through)
            tiles[0][lane * ld + r] =
                (lane < r && r < n) ? static_cast<ar_type>(raw00[i]) : ar_type{};
            if (two_tiles) {
                tiles[1][lane * ld + r] = (lane < r && B + r < n)
                                              ? static_cast<ar_type>(raw11[i])
                                              : ar_type{};
                tiles[2][lane * ld + r] =
                    (B + r < n) ? static_cast<ar_type>(raw10[i]) : ar_type{};
            }
        }
    }
    if (threadIdx.x == 0) {
        // `trsv_init` and the block counter of the reference
        // (cuda/trsv_kernels.cuh:38-42): nothing can expire here
        helper[0] = ~std::uint32_t{0};
        helper[1] = static_cast<std::uint32_t>(ceildiv(n, index_type{B}));
    }
    __syncthreads();
    if (wave != 0) {
        return;
    }
    // ---- the solve, wave 0 alone, no barrier from here on --------------------
    ar_type inv_diag[2] = {ar_type{1}, ar_type{1}};
    if (!UnitDiag) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h * B + lane < n) {
                inv_diag[h] = ar_type{1} / static_cast<ar_type>(diag_raw[h]);
            }
        }
    }
    const int rows0 = n < B ? static_cast<int>(n) : B;
    const ar_type r0 = lane < n ? static_cast<ar_type>(rhs_raw[0]) : ar_type{};
    const ar_type x0 = detail::substitute_tile64<UnitDiag>(
        tiles[0], 1, ld, r0, inv_diag[0], lane, rows0);
    const st_type x0_stored = static_cast<st_type>(x0);
    if (lane < n) {
        x_data[phys(lane) * x_inc] = x0_stored;
    }
    if (!two_tiles) {
        return;
    }
    // x_0..63 as later rows see them: as stored (`trsv_kernels.cuh:171-182`:
    // later blocks re-read x from memory)
    x_first[lane] = static_cast<ar_type>(x0_stored);
    lds_release();
    lds_acquire();
    // r1_i = rhs_i - sum_{j < 64} T10[i][j] x_j: four partial sums (j mod 4),
    // ascending j
    ar_type part[4] = {};
#pragma unroll
    for (int j = 0; j < B; ++j) {
        part[j & 3] += tiles[2][j * ld + lane] * x_first[j];
    }
    const ar_type r1 =
        B + lane < n ? static_cast<ar_type>(rhs_raw[1]) -
                           ((part[0] + part[1]) + (part[2] + part[3]))
                     : ar_type{};
    const ar_type x1 = detail::substitute_tile64<UnitDiag>(
        tiles[1], 1, ld, r1, inv_diag[1], lane, static_cast<int>(n) - B);
    if (B + lane < n) {
        x_data[phys(B + lane) * x_inc] = static_cast<st_type>(x1);
    }
}


}  // namespace kernel


/** Launches `kernel::trsv_small` (n <= `trsv_small_max_n`; any alignment, row
 *  stride and increment). */
template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv_small(bool upper, bool unit_diag, const MtxView &mtx,
                             const XView &x, std::uint32_t *helper,
                             hipStream_t stream)
{
    static_assert(view_traits<MtxView>::is_pointer_view &&
                      view_traits<XView>::is_pointer_view,
                  "reduced_row_major / row_major / plain views only");
    if (mtx.length(0) <= 0) {
        return hipSuccess;
    }
    if (mtx.length(0) > trsv_small_max_n) {
        return hipErrorInvalidValue;
    }
#define ACCBLAS_TRSV_SMALL_LAUNCH(_upper, _unit)         \
    kernel::trsv_small<_upper, _unit, ArType>            \
        <<<dim3(1), dim3(256), 0, stream>>>(mtx, x, helper)
    if (upper) {
        if (unit_diag) {
            ACCBLAS_TRSV_SMALL_LAUNCH(true, true);
        } else {
            ACCBLAS_TRSV_SMALL_LAUNCH(true, false);
        }
    } else {
        if (unit_diag) {
            ACCBLAS_TRSV_SMALL_LAUNCH(false, true);
        } else {
            ACCBLAS_TRSV_SMALL_LAUNCH(false, false);
        }
    }
#undef ACCBLAS_TRSV_SMALL_LAUNCH
    return hipGetLastError();
}


}  // namespace dev
}  // namespace accblas
