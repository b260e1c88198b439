// GEMV kernels for gfx950: res = alpha * A * x + beta * res, A row-major.
//
// Replaces `kernel::gemv` / `kernel::acc_gemv` and their launchers
// (`/root/reference/cuda/gemv_kernels.cuh:30-64,79-113,136-193`). The maths
// and the accessor contract are the reference's; the shape is not. The
// reference runs one 512-thread block per row with 4/8-byte scalar loads and
// a 4-barrier shared-memory tree. Here (MI355X, HBM-bound):
//   * a group of WPR waves owns R consecutive rows and sweeps them together;
//     every lane issues 16-byte loads of the STORAGE type (8 x fp16 /
//     4 x fp32 / 2 x fp64), so 8 lanes cover one 128 B line and one wave
//     instruction moves 1 KiB; R * UNROLL independent row loads per lane are
//     in flight before the first FMA;
//   * the x chunk a lane loads is widened once and reused for the R rows;
//   * values are widened to the arithmetic type in registers, FMA'd, reduced
//     with a 6-step 64-lane shuffle; waves of a group combine through LDS
//     (one value per wave and row), no barrier tree;
//   * power-of-two row pitches (the benchmark's 32768 fp32 columns) alias all
//     rows onto the same channel / bank bits; with fp64 arithmetic the rows of
//     a group are then skewed in time (`Skew`, see the kernel and launch_gemv);
//   * row pitches that are not a multiple of 128 bytes (the reference's sweep:
//     24500 fp32 columns) make every wave load straddle an extra cache line;
//     long rows then sweep per-row grids aligned to their own line boundaries
//     (`RowAlign`);
//   * a strided or unaligned x keeps the 16-byte matrix loads and fetches x
//     element-wise through `operator()` (VecX = false): x is L2-resident and
//     1/R of the matrix traffic, the stream stays the bound;
//   * anything the vector path cannot take (unaligned matrix base or row
//     stride) goes through `gemv_generic`, which uses only `operator()`.
#pragma once

#include <hip/hip_runtime.h>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


/** Tuning knobs of the vector kernel. */
template <int Waves, int WavesPerRowGroup, int RowsPerGroup, int Unroll,
          bool NonTemporal>
struct gemv_config {
    static constexpr int waves = Waves;          // waves per workgroup
    static constexpr int wpr = WavesPerRowGroup; // waves sharing a row group
    static constexpr int rows = RowsPerGroup;    // rows swept together
    static constexpr int unroll = Unroll;        // 16-byte loads per row/iter
    static constexpr bool nt = NonTemporal;
    static constexpr int groups = Waves / WavesPerRowGroup;
    static constexpr int rows_per_block = groups * RowsPerGroup;
    static_assert(Waves % WavesPerRowGroup == 0, "waves must split evenly");
};


namespace kernel {


/** Vec consecutive x entries starting at column c, widened to ArType: one
 *  16-byte load (VecX) or Vec accesses through the view (any increment). */
template <bool VecX, int Vec, typename XView, typename StType, typename ArType>
__device__ __forceinline__ void load_x(const XView &x, const StType *x_ptr,
                                       index_type c, ArType (&out)[Vec])
{
    if constexpr (VecX) {
        const auto xv = load_packed<Vec, false>(x_ptr + c);
#pragma unroll
        for (int k = 0; k < Vec; ++k) {
            out[k] = static_cast<ArType>(xv[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < Vec; ++k) {
            out[k] = x(c + k, 0);
        }
    }
}


/**
 * Vector-load GEMV. Requirements checked by the launcher: storage pointer of
 * mtx aligned to Vec*sizeof(St), row stride % Vec == 0; with VecX also x
 * aligned and of increment 1 (otherwise x goes through operator()).
 * res may have any increment (it is only touched through operator()).
 *
 * Follows `cuda/gemv_kernels.cuh:79-113` in meaning: local_result +=
 * mtx(row, col) * x(col, 0) in ArType, then
 * res(row,0) = alpha*sum (+ beta*res(row,0) if beta != 0).
 */
template <typename Config, int Vec, bool VecX, bool Skew, bool RowAlign,
          typename ArType, typename MtxView, typename XView, typename ResView>
__global__ __launch_bounds__(Config::waves *wave_size) void gemv_vec(
    ArType alpha, MtxView mtx, XView x, ArType beta, ResView res)
{
    using ar_type = decltype(alpha * mtx(0, 0) * x(0, 0) + beta * res(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename mtx_traits::storage_type>;
    constexpr int R = Config::rows;
    constexpr int U = Config::unroll;
    constexpr int WPR = Config::wpr;
    constexpr index_type wave_cols = index_type{wave_size} * Vec;
    constexpr index_type iter_cols = wave_cols * WPR * U;

    const int lane = lane_id();
    const int wave = wave_id();
    const int group = wave / WPR;
    const int wave_in_group = wave % WPR;

    const index_type num_rows = mtx.length(0);
    const index_type num_cols = mtx.length(1);
    const index_type row0 =
        (index_type{blockIdx.x} * Config::groups + group) * R;

    // Rows past the end are clamped to the last row for loading (keeps every
    // access in bounds and every wave on the barrier path) and skipped when
    // storing.
    const st_type *row_ptr[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const index_type r =
            (row0 + j < num_rows) ? row0 + j : num_rows - 1;
        row_ptr[j] = mtx_traits::data(mtx) + r * mtx_traits::stride(mtx);
    }
    const st_type *x_ptr = x_traits::data(x);

    ar_type acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        acc[j] = ar_type{};
    }

    // ---- main sweep: whole iterations, no bounds checks ------------------
    // (the row-aligned variant handles all columns itself)
    const index_type main_cols =
        RowAlign ? num_cols : (num_cols / iter_cols) * iter_cols;
    const index_type lane_off =
        index_type{wave_in_group} * wave_cols + index_type{lane} * Vec;
    if constexpr (RowAlign) {
        // Row pitch not a multiple of 128 bytes (the reference's sweep: 24500
        // fp32 columns = 98000 bytes): every row starts somewhere else inside
        // a cache line, and a sweep on a common column grid makes every 1 KiB
        // wave load straddle 9 lines instead of 8 (6.7 TB/s against 6.9-7.0
        // TB/s with 12 columns of padding). Here every row sweeps its OWN grid,
        // shifted to its next 128-byte boundary; x is then fetched per row
        // (from L2). The few columns before and after the aligned part are
        // done element-wise below.
        static_assert(VecX && !Skew, "row alignment needs a contiguous x");
        index_type delta[R];
        index_type max_delta = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const auto address = reinterpret_cast<std::uintptr_t>(row_ptr[j]);
            delta[j] = static_cast<index_type>((128 - (address & 127)) & 127) /
                       static_cast<index_type>(sizeof(st_type));
            max_delta = delta[j] > max_delta ? delta[j] : max_delta;
        }
        const index_type aligned_cols =
            num_cols > max_delta
                ? ((num_cols - max_delta) / iter_cols) * iter_cols
                : 0;
        for (index_type c0 = 0; c0 < aligned_cols; c0 += iter_cols) {
            ar_type xa[R][U][Vec];
            typename packed<st_type, Vec>::type av[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const index_type c =
                    c0 + index_type{u} * WPR * wave_cols + lane_off;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    av[j][u] = load_packed<Vec, Config::nt>(row_ptr[j] +
                                                            delta[j] + c);
                    load_x<true, Vec>(x, x_ptr, delta[j] + c, xa[j][u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < Vec; ++k) {
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        acc[j] +=
                            static_cast<ar_type>(av[j][u][k]) * xa[j][u][k];
                    }
                }
            }
        }
        // head [0, delta) and tail [delta + aligned_cols, cols) of every row
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const index_type r =
                (row0 + j < num_rows) ? row0 + j : num_rows - 1;
            const index_type head = delta[j] < num_cols ? delta[j] : num_cols;
            const index_type tail_begin = head + aligned_cols;
            const index_type count = head + (num_cols - tail_begin);
            for (index_type e = index_type{wave_in_group} * wave_size + lane;
                 e < count; e += WPR * wave_size) {
                const index_type cc = e < head ? e : tail_begin + (e - head);
                acc[j] += mtx(r, cc) * x(cc, 0);
            }
        }
    } else if constexpr (Skew) {
        // With a row pitch that is a multiple of a large power of two (the
        // benchmark's 32768 fp32 columns = 128 KiB) all rows have the same
        // channel and bank bits at a given column, and the loads a wave
        // issues together - and those of the groups that walk in step with it
        // - collide: 6.68 TB/s against 7.04-7.08 TB/s for the same matrix with
        // 128 bytes or 4 KiB more pitch (fp64 arithmetic; the shorter fp32
        // loop is hardly affected). So rows are skewed in time: in round t row
        // j of a group sweeps chunk (phase + t + j) mod T, with a phase per
        // group. Every x chunk is still loaded once per wave: a window of R
        // chunks stays in registers, slot = position mod R. The launcher
        // guarantees T % R == 0 and T >= R.
        static_assert(VecX && R > 1, "skew needs a contiguous x and R > 1");
        const index_type num_iters = main_cols / iter_cols;
        // row groups additionally start at different chunks (5 is coprime
        // with the power-of-two chunk counts this path is used for)
        const index_type phase =
            ((index_type{blockIdx.x} * Config::groups + group) * 5) % num_iters;
        const auto wrap = [num_iters](index_type chunk) {
            chunk -= chunk >= num_iters ? num_iters : 0;
            chunk -= chunk >= num_iters ? num_iters : 0;
            return chunk;
        };
        ar_type window[R][U][Vec];
        const auto load_window = [&](index_type chunk, auto &slot) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const index_type c = chunk * iter_cols +
                                     index_type{u} * WPR * wave_cols + lane_off;
                load_x<true, Vec>(x, x_ptr, c, slot[u]);
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            load_window(wrap(phase + s), window[s]);
        }
        for (index_type t0 = 0; t0 < num_iters; t0 += R) {
#pragma unroll
            for (int p = 0; p < R; ++p) {
                const index_type t = t0 + p;
                // chunk t + R - 1 enters the slot chunk t - 1 left
                load_window(wrap(phase + t + R - 1), window[(p + R - 1) % R]);
                typename packed<st_type, Vec>::type av[R][U];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const index_type chunk = wrap(phase + t + j);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        av[j][u] = load_packed<Vec, Config::nt>(
                            row_ptr[j] + chunk * iter_cols +
                            index_type{u} * WPR * wave_cols + lane_off);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < Vec; ++k) {
#pragma unroll
                        for (int j = 0; j < R; ++j) {
                            acc[j] += static_cast<ar_type>(av[j][u][k]) *
                                      window[(p + j) % R][u][k];
                        }
                    }
                }
            }
        }
    } else {
        for (index_type c0 = 0; c0 < main_cols; c0 += iter_cols) {
            ar_type xa[U][Vec];
            typename packed<st_type, Vec>::type av[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const index_type c =
                    c0 + index_type{u} * WPR * wave_cols + lane_off;
                load_x<VecX, Vec>(x, x_ptr, c, xa[u]);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    av[j][u] = load_packed<Vec, Config::nt>(row_ptr[j] + c);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < Vec; ++k) {
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        acc[j] += static_cast<ar_type>(av[j][u][k]) * xa[u][k];
                    }
                }
            }
        }
    }

    // ---- ragged end: the whole vectors left of every row in ONE batch of
    // loads (they are fewer than U per lane: one memory round trip instead of
    // one per vector - short rows consist of little else, and a row's first
    // access often pays a page walk on top), then the last partial vector
    // element-wise. Lanes without a vector load column 0 and add nothing.
    {
        const index_type whole_cols = (num_cols / Vec) * Vec;
        ar_type xa[U][Vec];
        typename packed<st_type, Vec>::type av[R][U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const index_type c =
                main_cols + index_type{u} * WPR * wave_cols + lane_off;
            live[u] = !RowAlign && c + Vec <= whole_cols;
            const index_type cc = live[u] ? c : 0;
            // (a row shorter than one vector has no column 0 to fall back on)
            if (!RowAlign && whole_cols >= Vec) {
                load_x<VecX, Vec>(x, x_ptr, cc, xa[u]);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    av[j][u] = load_packed<Vec, Config::nt>(row_ptr[j] + cc);
                }
            }
        }
        if constexpr (!RowAlign) {
            if (whole_cols >= Vec) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < Vec; ++k) {
#pragma unroll
                        for (int j = 0; j < R; ++j) {
                            const ar_type term =
                                static_cast<ar_type>(av[j][u][k]) * xa[u][k];
                            acc[j] += live[u] ? term : ar_type{};
                        }
                    }
                }
            }
            // fewer than Vec columns are left: one lane each
            const index_type cc = whole_cols + lane;
            if (wave_in_group == 0 && cc < num_cols) {
                const ar_type xk = x(cc, 0);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const index_type r =
                        (row0 + j < num_rows) ? row0 + j : num_rows - 1;
                    acc[j] += mtx(r, cc) * xk;
                }
            }
        }
    }

    // ---- reduce: lanes by shuffle, waves of a group through LDS ----------
    __shared__ ar_type scratch[Config::groups * R * WPR];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        acc[j] = wave_reduce_sum(acc[j]);
    }
    if (WPR > 1) {
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                scratch[(group * R + j) * WPR + wave_in_group] = acc[j];
            }
        }
        __syncthreads();
        // lane j of the group's first wave finishes row j
        if (wave_in_group == 0 && lane < R) {
            ar_type total{};
#pragma unroll
            for (int w = 0; w < WPR; ++w) {
                total += scratch[(group * R + lane) * WPR + w];
            }
            const index_type row = row0 + lane;
            if (row < num_rows) {
                if (beta == ArType{0}) {
                    res(row, 0) = alpha * total;
                } else {
                    res(row, 0) = alpha * total + beta * res(row, 0);
                }
            }
        }
    } else {
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const index_type row = row0 + j;
                if (row < num_rows) {
                    if (beta == ArType{0}) {
                        res(row, 0) = alpha * acc[j];
                    } else {
                        res(row, 0) = alpha * acc[j] + beta * res(row, 0);
                    }
                }
            }
        }
    }
}


/**
 * Fallback for any stride / alignment: one wave per row, lanes stride the
 * columns, everything through `operator()` (4 independent loads per lane in
 * flight). Same meaning as `cuda/gemv_kernels.cuh:79-113`.
 */
template <int Waves, typename ArType, typename MtxView, typename XView,
          typename ResView>
__global__ __launch_bounds__(Waves *wave_size) void gemv_generic(
    ArType alpha, MtxView mtx, XView x, ArType beta, ResView res)
{
    using ar_type = decltype(alpha * mtx(0, 0) * x(0, 0) + beta * res(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    constexpr int U = 4;
    const int lane = lane_id();
    const index_type row = index_type{blockIdx.x} * Waves + wave_id();
    const index_type num_cols = mtx.length(1);
    if (row >= mtx.length(0)) {
        return;
    }
    ar_type partial[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        partial[u] = ar_type{};
    }
    index_type col = lane;
    for (; col + (U - 1) * wave_size < num_cols; col += U * wave_size) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            partial[u] +=
                mtx(row, col + u * wave_size) * x(col + u * wave_size, 0);
        }
    }
    for (; col < num_cols; col += wave_size) {
        partial[0] += mtx(row, col) * x(col, 0);
    }
    ar_type sum = (partial[0] + partial[1]) + (partial[2] + partial[3]);
    sum = wave_reduce_sum(sum);
    if (lane == 0) {
        if (beta == ArType{0}) {
            res(row, 0) = alpha * sum;
        } else {
            res(row, 0) = alpha * sum + beta * res(row, 0);
        }
    }
}


}  // namespace kernel


/** Production shape of the vector kernel, chosen by measurement on MI355X
 *  (tools/tune_gemv.hip, profiles/r01_tune_gemv*.txt): 8 waves in two groups of
 *  4, each group sweeps 4 rows together, 2 x 16-byte loads per row and lane in
 *  flight, non-temporal matrix loads (the matrix is read once: nt gained 12 %
 *  over default-policy loads at 32768^2, 6.89 vs 6.17 TB/s). Every nt shape
 *  with 8-16 loads in flight per lane lands within 2 % of this one. */
using gemv_default_config = gemv_config<8, 4, 4, 2, true>;

/** Shape of the time-skewed variant used for power-of-two row pitches. */
/** Shape of the time-skewed variant (power-of-two row pitch, fp64
 *  arithmetic): four groups of two waves, two rows each; measured against
 *  eleven other shapes in profiles/r01_gemv_skew_ab.txt. */
using gemv_skew_config = gemv_config<8, 2, 2, 2, true>;

/** Shape of the variant for row pitches that are not a multiple of 128 bytes
 *  (every row sweeps its own cache-line-aligned column grid): two groups of
 *  four waves, four rows each, one 16-byte load per row and step; measured
 *  against nine other shapes in profiles/r01_gemv_row_align_ab.txt. */
using gemv_row_align_config = gemv_config<8, 4, 4, 1, true>;

/** Short rows (at most `gemv_short_row_bytes<St>` of storage per row: the first
 *  quarter of the reference's sweep, cuda/gemv_benchmark.cu:211-214, and tall
 *  matrices): the production shape covers 8 KiB of a row per step and would run
 *  such rows through its ragged end only, with 8 rows per workgroup a small
 *  matrix leaves half of the CUs without work (n = 1300: 163 workgroups on 256
 *  CUs). Here two waves share a row and cover 4 KiB per step. Few rows: one row
 *  per 128-thread workgroup, so that even n = 1100 spreads over all CUs; many
 *  rows: two groups of two waves, two rows each (x re-used for two rows).
 *  Both have the SAME summation order per row (it depends on the waves per
 *  row and the unroll only), so a row block gives the same bits alone as
 *  inside a taller product. Measured against 20 other shapes and the rocBLAS
 *  columns in profiles/r03_tune_gemv_small.txt (lda = 24500, us, production
 *  shape -> this one | rocBLAS; before the ragged end of the kernel became one
 *  batch of loads, which closed most of the gap for square matrices):
 *  Acc<fp64,fp32> n = 1300 6.80 -> 6.24 | 6.28, 1900 8.04 -> 6.24 | 6.68,
 *  2200 9.84 -> 6.56 | 8.32, 3000 11.4 -> 9.5 | 13.7, 4500 21.2 -> 17.2 | 23.7;
 *  fp64 2200 13.4 -> 9.5 | 12.8; 24000 x 2000 Acc<fp64,fp32> 40.1 -> 33.5 |
 *  52.0.
 *  How long "short" is depends on the storage type. The production shape has 8
 *  rows per workgroup: a CU gets ceil(rows / 8 / 256) of them, and just past a
 *  multiple of 2048 rows that is one more for a few CUs - n = 4100 takes 25 %
 *  longer than n = 4000, n = 6200 32 % longer than 6000 - while the matrix
 *  still streams faster than HBM (Infinity Cache) and the balance between the
 *  CUs is what counts. The short shapes have 1 or 4 rows per workgroup. Measured
 *  (profiles/r03_tune_gemv_mid.txt, lda = 24500, us, production shape -> best
 *  short shape | rocBLAS): fp64 n = 3100 19.4 -> 16.7 | 17.9, 3500 21.2 -> 19.3
 *  | 19.7, 4000 24.2 -> 24.2 | 23.7, 4100 30.2 -> 25.8 | 27.9, 4500 33.8 ->
 *  29.8 | 32.0, 5000 37.8 -> 34.2 | 38.0, 5200 40.5 -> 40.5, 5500 42.9 -> 43.6;
 *  Acc<fp64,fp32> 6200 34.7 -> 30.8 | 38.0, 6500 35.9 -> 32.3, 7000 38.1 ->
 *  34.9, 8000 43.3 -> 45.4. So: 40 KiB per row with 8-byte storage, 28 KiB with
 *  4-byte storage, 24 KiB with 2-byte storage (not measured beyond). Tall
 *  matrices of such rows lose 2 % (24000 x 4000 fp64: 120.9 -> 123.2). */
using gemv_short_config = gemv_config<2, 2, 1, 2, true>;
using gemv_short_tall_config = gemv_config<4, 2, 2, 2, true>;
template <typename StorageType>
constexpr index_type gemv_short_row_bytes =
    sizeof(StorageType) == 8
        ? index_type{40} << 10
        : (sizeof(StorageType) == 4 ? index_type{28} << 10
                                    : index_type{24} << 10);
constexpr index_type gemv_short_max_rows = 4352;  // above: the two-row shape
static_assert(gemv_short_config::wpr == gemv_short_tall_config::wpr &&
                  gemv_short_config::unroll == gemv_short_tall_config::unroll,
              "the two short-row shapes must sum a row in the same order");

/** Matrices that fit the Infinity Cache (256 MiB; with the way it is filled a
 *  cyclic re-read still hits up to ~320 MB) are loaded with the DEFAULT cache
 *  policy instead of the non-temporal hint: an iterative solver - what the
 *  accessor is for - multiplies with the same matrix again and again, and so
 *  does the reference's benchmark (10 timed runs per size). Measured against the
 *  non-temporal production shape (profiles/r04_tune_gemv_mall.txt, lda = 24500,
 *  us): Acc<fp64,fp32> n = 5500 24.0 -> 21.2, 6500 32.3 -> 28.0, 7000 35.0 ->
 *  30.8, 8000 42.4 -> 38.7, 8500 50.9 -> 46.0, 9000 54.8 -> 52.6, but 10000
 *  (400 MB) 63.4 -> 67.8; fp64 4000 24.8 -> 21.6, 5000 34.7 -> 30.9, 6000 46.8
 *  -> 44.0, equal from 6500 (338 MB) on. The cached shape has two rows per
 *  group instead of four (finer balance between the CUs while the data comes
 *  from the cache) and the SAME waves per row and unroll as the production
 *  shape: a row is summed in the same order, so a row block still has the bits
 *  of the same rows inside a larger product whichever of the two it runs. */
using gemv_cached_config = gemv_config<8, 4, 2, 2, false>;
constexpr index_type gemv_cached_max_bytes = index_type{320} * 1000 * 1000;
/** The same for SHORT rows (`gemv_short_row_bytes`): the two-row short shape
 *  without the hint, from 80 MB on (below, the non-temporal short shapes are as
 *  fast or faster: the whole launch is a few microseconds). Same table, us,
 *  library (short shapes, non-temporal) -> this one: Acc<fp64,fp32> n = 4500 17.7
 *  -> 17.1, 5500 24.2 -> 22.4, 6500 32.8 -> 29.7, 7000 34.8 -> 32.3; fp64 3500
 *  18.9 -> 18.4, 4000 23.7 -> 21.6 (rocBLAS 23.8), 4500 29.9 -> 26.4, 5000 34.2
 *  -> 30.9. Waves per row and unroll of the short shapes: the same bits. */
using gemv_short_cached_config = gemv_config<4, 2, 2, 2, false>;
constexpr index_type gemv_cached_min_bytes = index_type{80} * 1000 * 1000;
static_assert(gemv_cached_config::wpr == gemv_default_config::wpr &&
                  gemv_cached_config::unroll == gemv_default_config::unroll,
              "the cached shape must sum a row in the production shape's order");
static_assert(gemv_short_cached_config::wpr == gemv_short_config::wpr &&
                  gemv_short_cached_config::unroll == gemv_short_config::unroll,
              "the cached short shape must sum a row in the short shapes' order");

/** When the time-skewed sweep is used (option "gemv_skew" of the C ABI). */
enum class gemv_skew_mode {
    never = 0,
    automatic = 1,  // fp64 arithmetic and a power-of-two row pitch (see launch_gemv)
    always = 2      // whenever the shape allows it (measurements)
};


/**
 * Picks the widest vector path the operands allow and launches. Returns the
 * hipError_t of the launch.
 */
template <typename Config = gemv_default_config, typename ArType,
          typename MtxView, typename XView, typename ResView>
hipError_t launch_gemv(ArType alpha, const MtxView &mtx, const XView &x,
                       ArType beta, const ResView &res, hipStream_t stream,
                       gemv_skew_mode skew_mode = gemv_skew_mode::automatic,
                       bool any_alignment = true)
{
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename mtx_traits::storage_type>;
    constexpr int vec16 = 16 / sizeof(st_type);
    constexpr int vec8 = 8 / sizeof(st_type);

    const index_type rows = mtx.length(0);
    if (rows <= 0) {
        return hipSuccess;
    }
    const auto *a_ptr = mtx_traits::data(mtx);
    const auto *x_ptr = x_traits::data(x);
    const index_type lda = mtx_traits::stride(mtx);
    const bool unit_x = x_traits::stride(x) == 1 || x.length(0) <= 1;

    const dim3 block(Config::waves * wave_size, 1, 1);
    const dim3 grid(
        static_cast<unsigned>(ceildiv(rows, Config::rows_per_block)), 1, 1);
    // accessor kinds whose elements are not plain storage (scaled, ...) only
    // exist through operator()
    constexpr bool pointer_views =
        mtx_traits::is_pointer_view && x_traits::is_pointer_view;
    const auto launch_generic = [&]() {
        constexpr int waves = 4;
        kernel::gemv_generic<waves>
            <<<dim3(static_cast<unsigned>(ceildiv(rows, waves))),
               dim3(waves * wave_size), 0, stream>>>(alpha, mtx, x, beta, res);
    };
    if constexpr (!pointer_views) {
        launch_generic();
    } else {
        // 16-byte loads need no alignment (`load_packed`): rows and x are taken
        // as they come. Until round 3 (`any_alignment` = false) operands that
        // were not 16-byte aligned ran with 8-byte loads or element-wise: an odd
        // row stride cost 25 % (fp32 storage) to 35 % (fp64) at n = 24501
        // (profiles/r04_gemv_odd_strides.txt).
        const bool a_aligned = lda % vec16 == 0 && is_aligned(a_ptr, 16);
        const bool x_vectors =
            unit_x && (is_aligned(x_ptr, 16) ||
                       (any_alignment && is_aligned(x_ptr, sizeof(st_type))));
        // (short unaligned rows of 2- and 4-byte storage stay with the
        // element-wise kernel: 2049 / 4097 columns 6.5 / 14.0 us against 7.4 /
        // 15.3 with vectors across cache lines, 8001 columns 40.1 against 37.9)
        const bool a_vectors =
            a_aligned ||
            (any_alignment && x_vectors && is_aligned(a_ptr, sizeof(st_type)) &&
             (sizeof(st_type) == 8 || mtx.length(1) >= 6144));
        if (a_vectors) {
            if (x_vectors) {
                // power-of-two row pitch and fp64 arithmetic: time-skewed
                // rows (see the kernel); with fp32 arithmetic the plain sweep
                // is the faster one (612 vs 622 us at 32768^2)
                using skew = gemv_skew_config;
                constexpr index_type skew_iter_cols =
                    index_type{wave_size} * vec16 * skew::wpr * skew::unroll;
                constexpr index_type row_align_iter_cols =
                    index_type{wave_size} * vec16 * gemv_row_align_config::wpr *
                    gemv_row_align_config::unroll;
                const index_type cols = mtx.length(1);
                const index_type iters = cols / skew_iter_cols;
                const index_type pitch_bytes =
                    lda * static_cast<index_type>(sizeof(st_type));
                // measured (profiles/r01_gemv_skew_ab.txt): it pays for fp64
                // arithmetic on power-of-two pitches from 16 KiB on (-16 % at
                // 32 KiB ... -4 % at 256 KiB with fp32 storage), not for
                // pitches like 96 or 192 KiB, not for fp32 arithmetic, and
                // not for fp64 storage beyond 128 KiB
                const bool power_of_two =
                    (pitch_bytes & (pitch_bytes - 1)) == 0;
                const bool wanted =
                    skew_mode == gemv_skew_mode::always ||
                    (skew_mode == gemv_skew_mode::automatic &&
                     std::is_same<ArType, double>::value && power_of_two &&
                     pitch_bytes >= (index_type{16} << 10) &&
                     (sizeof(st_type) < 8 ||
                      pitch_bytes <= (index_type{128} << 10)));
                if (wanted && iters >= skew::rows &&
                    iters % skew::rows == 0) {
                    const dim3 skew_grid(
                        static_cast<unsigned>(
                            ceildiv(rows, skew::rows_per_block)),
                        1, 1);
                    kernel::gemv_vec<skew, vec16, true, true, false>
                        <<<skew_grid, dim3(skew::waves * wave_size), 0,
                           stream>>>(alpha, mtx, x, beta, res);
                } else if ((pitch_bytes % 128 != 0 || !a_aligned) &&
                           cols >= 12 * row_align_iter_cols) {
                    // rows start inside cache lines: per-row aligned grids
                    // (24500^2 fp32: 357 -> 339 us; below ~12 steps per row
                    // the element-wise head and tail cost more than the
                    // alignment gains). Rows that are not even 16-byte aligned
                    // get aligned matrix loads this way as well; only x is
                    // then read across its 16-byte boundaries.
                    using aligned = gemv_row_align_config;
                    const dim3 aligned_grid(
                        static_cast<unsigned>(
                            ceildiv(rows, aligned::rows_per_block)),
                        1, 1);
                    kernel::gemv_vec<aligned, vec16, true, false, true>
                        <<<aligned_grid, dim3(aligned::waves * wave_size), 0,
                           stream>>>(alpha, mtx, x, beta, res);
                } else if (std::is_same<Config, gemv_default_config>::value &&
                           cols * static_cast<index_type>(sizeof(st_type)) <=
                               gemv_short_row_bytes<st_type>) {
                    // short rows (the choice between the shapes does not
                    // change the bits of a row, see gemv_short_config)
                    const index_type matrix_bytes =
                        rows * cols * static_cast<index_type>(sizeof(st_type));
                    if (matrix_bytes >= gemv_cached_min_bytes &&
                        matrix_bytes <= gemv_cached_max_bytes) {
                        // fits the Infinity Cache and is large enough for that
                        // to matter: the two-row shape with the default cache
                        // policy (see gemv_cached_config)
                        using shape = gemv_short_cached_config;
                        kernel::gemv_vec<shape, vec16, true, false, false>
                            <<<dim3(static_cast<unsigned>(
                                   ceildiv(rows, shape::rows_per_block))),
                               dim3(shape::waves * wave_size), 0, stream>>>(
                                alpha, mtx, x, beta, res);
                    } else if (rows <= gemv_short_max_rows) {
                        using shape = gemv_short_config;
                        kernel::gemv_vec<shape, vec16, true, false, false>
                            <<<dim3(static_cast<unsigned>(
                                   ceildiv(rows, shape::rows_per_block))),
                               dim3(shape::waves * wave_size), 0, stream>>>(
                                alpha, mtx, x, beta, res);
                    } else {
                        using shape = gemv_short_tall_config;
                        kernel::gemv_vec<shape, vec16, true, false, false>
                            <<<dim3(static_cast<unsigned>(
                                   ceildiv(rows, shape::rows_per_block))),
                               dim3(shape::waves * wave_size), 0, stream>>>(
                                alpha, mtx, x, beta, res);
                    }
                } else if (std::is_same<Config, gemv_default_config>::value &&
                           rows * cols *
                                   static_cast<index_type>(sizeof(st_type)) <=
                               gemv_cached_max_bytes) {
                    // fits the Infinity Cache: default cache policy
                    using shape = gemv_cached_config;
                    kernel::gemv_vec<shape, vec16, true, false, false>
                        <<<dim3(static_cast<unsigned>(
                               ceildiv(rows, shape::rows_per_block))),
                           dim3(shape::waves * wave_size), 0, stream>>>(
                            alpha, mtx, x, beta, res);
                } else {
                    kernel::gemv_vec<Config, vec16, true, false, false>
                        <<<grid, block, 0, stream>>>(alpha, mtx, x, beta, res);
                }
            } else {
                kernel::gemv_vec<Config, vec16, false, false, false>
                    <<<grid, block, 0, stream>>>(alpha, mtx, x, beta, res);
            }
        } else if (vec8 > 1 && lda % vec8 == 0 && is_aligned(a_ptr, 8)) {
            if (unit_x && is_aligned(x_ptr, 8)) {
                kernel::gemv_vec<Config, (vec8 > 1 ? vec8 : 2), true, false,
                                 false>
                    <<<grid, block, 0, stream>>>(alpha, mtx, x, beta, res);
            } else {
                kernel::gemv_vec<Config, (vec8 > 1 ? vec8 : 2), false, false,
                                 false>
                    <<<grid, block, 0, stream>>>(alpha, mtx, x, beta, res);
            }
        } else {
            launch_generic();
        }
    }
    return hipGetLastError();
}


}  // namespace dev
}  // namespace accblas
