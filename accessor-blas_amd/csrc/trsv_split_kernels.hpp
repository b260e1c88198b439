// Role-split TRSV for large systems on gfx950.
//
// Same maths, hand-off protocol and accessor contract as `kernel::trsv`
// (trsv_kernels.hpp; reference: `/root/reference/cuda/trsv_kernels.cuh:527-692,
// 725-893`), different work distribution. In `kernel::trsv` the workgroup that
// solves row block b also streams all of b's off-diagonal panels. Only a few
// hundred workgroups are resident, tickets are handed out in row order, so a
// late block (b ~ 1000 at n = 65536) has ~16 MB to stream but lives on ONE
// compute unit for ~256 hops of the front: the solve ends up bound by per-CU
// bandwidth (~25 GB/s) with the other CUs idle early on (measured 3.6 TB/s).
//
// Here ONE launch contains two roles, assigned by ticket (start order):
//   * the first G workgroups to start are STREAMERS. Streamer s owns the row
//     blocks {s, s+G, s+2G, ...} (cyclic => balanced at every moment), keeps
//     their partial sums in registers and walks the column panels c = 0, 1, ...
//     in order: as soon as x_c is published it applies panel (b, c) for every
//     owned block b >= c + 3 (16-byte loads, double-buffered one panel ahead),
//     and when block b has received its last far panel (c = b - 3) it reduces
//     and publishes the 64 far corrections of b as granules. Streamers are
//     persistent, hence resident for the whole solve: bandwidth is used from
//     the first hop on, by all CUs.
//   * every later workgroup is the SOLVER of one row block (ticket order =
//     row order, as in the reference): it stages the diagonal tile and the TWO
//     panels next to it, inverts, folds (M2 = inv(T) A_{b,b-2} in registers,
//     M1 = inv(T) A_{b,b-1} in LDS), then waits for its far corrections and
//     x_{b-2} (both known at least one hop early), then for x_{b-1}, and
//     publishes x_b. It streams nothing.
// A workgroup only ever waits on workgroups that hold an earlier ticket
// (streamers hold the first G; solver b waits on streamer (b mod G) and on
// solver b-1), so the scheme is deadlock-free as long as at least G + 1
// workgroups fit on the device, which the launcher checks with the occupancy
// query (it uses at most half of the reported capacity for streamers).
#pragma once

#include <hip/hip_runtime.h>

#include "trsv_kernels.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_split_waves = 4;       // waves per workgroup (both roles)
constexpr int trsv_split_max_owned = 4;   // row blocks per streamer


namespace kernel {


/**
 * blockDim = 256, grid = G + ceil(n / 64). `x_granules` holds one entry per
 * element of x (storage type), `corr_granules` one entry per row (arithmetic
 * type); both zeroed by `trsv_init`. Requirements (checked by the launcher):
 * Vec > 1 path operands (aligned base / stride, Upper: n % Vec == 0),
 * ceil(n/64) <= trsv_split_max_owned * G.
 */
template <bool Upper, bool UnitDiag, int Vec, typename ArType,
          typename MtxView, typename XView>
__global__ __launch_bounds__(trsv_split_waves *wave_size, 2) void trsv_split(
    MtxView mtx, XView x, std::uint32_t *helper, granule_word *x_granules,
    granule_word *corr_granules, int num_streamers)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    using x_io = detail::granule_io<st_type>;
    using c_io = detail::granule_io<ar_type>;
    constexpr int Waves = trsv_split_waves;
    constexpr int B = trsv_tile;
    constexpr int ld = B + 1;
    constexpr int RW = B / Waves;   // rows of a block per streamer wave
    constexpr int LPR = B / Vec;    // lanes sharing one tile row
    constexpr int NI = RW / Vec;    // load instructions per panel and block
    constexpr int CW = B / Waves;   // columns per solver wave in the mat-vecs
    constexpr int MO = trsv_split_max_owned;
    static_assert(Vec > 1 && RW % Vec == 0, "vector path only");

    __shared__ ar_type tile[B * ld];
    __shared__ ar_type panel[B * ld];
    __shared__ ar_type correction[B];
    __shared__ ar_type partial_a[Waves * B];
    __shared__ ar_type partial_b[Waves * B];
    __shared__ ar_type x_prev_lds[B];
    __shared__ ar_type x_prev2_lds[B];
    __shared__ std::int32_t shared_ticket;

    const int lane = lane_id();
    const int wave = wave_id();
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const index_type n = mtx.length(0);
    const index_type num_blocks = ceildiv(n, B);

    if (threadIdx.x == 0) {
        shared_ticket = static_cast<std::int32_t>(
            __hip_atomic_fetch_add(helper + 1, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    const index_type ticket = shared_ticket;
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };
    const st_type *a_data = mtx_traits::data(mtx);
    const index_type lda = mtx_traits::stride(mtx);
    unsigned int spins = 0;

    if (ticket < num_streamers) {
        // =================================================== STREAMER ======
        const index_type first = ticket;
        const int row_in_inst = lane / LPR;
        const int col_group = lane % LPR;
        // owned blocks first + o * G; only blocks >= 3 have far panels (the
        // solver itself applies the two panels next to the diagonal)
        index_type block_of[MO];
        index_type last_owned = -1;
#pragma unroll
        for (int o = 0; o < MO; ++o) {
            block_of[o] = first + index_type{o} * num_streamers;
            if (block_of[o] < num_blocks && block_of[o] >= 3) {
                last_owned = block_of[o];
            }
        }
        if (last_owned < 0) {
            return;
        }
        const index_type panel_end = last_owned - 2;  // panels [0, b_max - 3]
        ar_type acc[MO][NI];
#pragma unroll
        for (int o = 0; o < MO; ++o) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                acc[o][k] = ar_type{};
            }
        }
        struct stage {
            st_type a[MO][NI][Vec];
            granule_word xraw[Vec][x_io::words];
        };
        // block o takes part in panel c iff it exists and c <= block - 3
        const auto active = [&](int o, index_type c) {
            return block_of[o] < num_blocks && block_of[o] >= c + 3;
        };
        const auto issue = [&](index_type c, stage &s) {
            const index_type first_col = c * B + col_group * Vec;  // logical
            const index_type mem_col =
                Upper ? n - 1 - (first_col + Vec - 1) : first_col;
#pragma unroll
            for (int o = 0; o < MO; ++o) {
                if (active(o, c)) {
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        const index_type gr = block_of[o] * B + wave_u * RW +
                                              k * Vec + row_in_inst;
                        const st_type *ptr =
                            a_data + phys(gr < n ? gr : n - 1) * lda + mem_col;
                        const auto v = load_packed<Vec, true>(ptr);
#pragma unroll
                        for (int e = 0; e < Vec; ++e) {
                            s.a[o][k][e] = v[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                x_io::load(x_granules, first_col + e, s.xraw[e]);
            }
        };
        const auto consume = [&](index_type c, stage &s) {
            const index_type first_col = c * B + col_group * Vec;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < Vec; ++e) {
                    ok = ok && x_io::ready(s.xraw[e]);
                }
                if (__all(ok) || detail::spin_expired(spins, helper)) {
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int e = 0; e < Vec; ++e) {
                    x_io::load(x_granules, first_col + e, s.xraw[e]);
                }
            }
            ar_type xv[Vec];
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                xv[e] = static_cast<ar_type>(x_io::decode(s.xraw[e]));
            }
#pragma unroll
            for (int o = 0; o < MO; ++o) {
                if (active(o, c)) {
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
#pragma unroll
                        for (int e = 0; e < Vec; ++e) {
                            acc[o][k] += static_cast<ar_type>(s.a[o][k][e]) *
                                         xv[Upper ? Vec - 1 - e : e];
                        }
                    }
                    if (block_of[o] == c + 3) {
                        // last far panel of this block: publish corrections
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            const ar_type total = group_all_reduce<LPR>(
                                acc[o][k],
                                [](ar_type a, ar_type b) { return a + b; });
                            if (col_group == 0) {
                                const index_type row = block_of[o] * B +
                                                       wave_u * RW + k * Vec +
                                                       row_in_inst;
                                c_io::publish(corr_granules, row,
                                              row < n ? total : ar_type{});
                            }
                        }
                    }
                }
            }
        };
        stage even, odd;
        issue(0, even);
        index_type c = 0;
        for (; c + 1 < panel_end; c += 2) {
            issue(c + 1, odd);
            consume(c, even);
            if (c + 2 < panel_end) {
                issue(c + 2, even);
            }
            consume(c + 1, odd);
        }
        if (c < panel_end) {
            consume(c, even);
        }
        return;
    }

    // ======================================================= SOLVER ========
    const index_type block = ticket - num_streamers;
    const index_type base = block * B;
    if (base >= n) {
        return;
    }
    st_type *x_data = x_traits::data(x);
    const index_type x_inc = x_traits::stride(x);
    const index_type own_row = base + lane;
    const bool own_valid = own_row < n;
    st_type *own_addr = x_data + phys(own_valid ? own_row : n - 1) * x_inc;
    ar_type own_rhs{};
    if (own_valid) {
        own_rhs = static_cast<ar_type>(*own_addr);
    }

    // stages the 64 x 64 panel of logical block column `cb` into `panel`
    // (column-major) and the diagonal tile into `tile` (row-major) if asked
    const auto stage_panel = [&](index_type cb) {
        for (int r = wave; r < B; r += Waves) {
            const index_type gr = base + r;
            ar_type off{};
            if (gr < n) {
                off = mtx(phys(gr), phys(cb * B + lane));
            }
            panel[lane * ld + r] = off;
        }
    };
    for (int r = wave; r < B; r += Waves) {
        const index_type gr = base + r;
        const index_type gc = base + lane;
        ar_type value{};
        if (lane == r) {
            value = (UnitDiag || gr >= n) ? ar_type{1}
                                          : ar_type(mtx(phys(gr), phys(gc)));
        } else if (lane < r && gr < n) {
            value = mtx(phys(gr), phys(gc));
        }
        tile[r * ld + lane] = value;
    }
    if (block >= 2) {
        stage_panel(block - 2);
    }
    __syncthreads();

    // invert the tile: lane j computes column j of the inverse
    if (wave == 0) {
        ar_type column[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            ar_type sums[4] = {};
#pragma unroll
            for (int k = 0; k < i; ++k) {
                sums[k & 3] += tile[i * ld + k] * column[k];
            }
            ar_type value = ((lane == i) ? ar_type{1} : ar_type{0}) -
                            ((sums[0] + sums[1]) + (sums[2] + sums[3]));
            if (!UnitDiag) {
                value = value / tile[i * ld + i];
            }
            column[i] = value;
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            tile[lane * ld + i] = column[i];
        }
    }
    __syncthreads();

    // fold the inverse into the two panels next to the diagonal:
    //   x_b = inv(T) (rhs_far) - M2 x_{b-2} - M1 x_{b-1},
    //   M2 = inv(T) A_{b,b-2} kept in REGISTERS (CW entries of this thread's
    //   row), M1 = inv(T) A_{b,b-1} kept in LDS in place of the panel.
    // Two near panels give the streamers two hops of lead for the far
    // corrections, which keeps their hand-off off the chain's 2-hop path.
    ar_type m2[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        m2[j] = ar_type{};
    }
    if (block > 0) {
        ar_type inv_row[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            inv_row[k] = tile[k * ld + lane];
        }
        constexpr int CG = 4;
        if (block >= 2) {
#pragma unroll
            for (int c0 = 0; c0 < CW; c0 += CG) {
                const int col = wave_u * CW + c0;
#pragma unroll
                for (int k = 0; k < B; ++k) {
#pragma unroll
                    for (int q = 0; q < CG; ++q) {
                        m2[c0 + q] += inv_row[k] * panel[(col + q) * ld + k];
                    }
                }
            }
            __syncthreads();  // every wave is done reading panel b-2
        }
        stage_panel(block - 1);
        __syncthreads();
        for (int c0 = 0; c0 < CW; c0 += CG) {
            const int col = wave_u * CW + c0;
            ar_type m[CG];
#pragma unroll
            for (int q = 0; q < CG; ++q) {
                m[q] = ar_type{};
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    m[q] += inv_row[k] * panel[(col + q) * ld + k];
                }
            }
#pragma unroll
            for (int q = 0; q < CG; ++q) {
                panel[(col + q) * ld + lane] = m[q];
            }
        }
    }

    // far corrections from this block's streamer (blocks 0..2 have none) are
    // polled by wave 1, x_{b-2} by wave 0; both arrive at least one hop
    // before x_{b-1}
    if (wave == 1) {
        ar_type far{};
        if (block >= 3) {
            granule_word raw[c_io::words];
            c_io::load(corr_granules, base + lane, raw);
            while (!__all(c_io::ready(raw)) &&
                   !detail::spin_expired(spins, helper)) {
                __builtin_amdgcn_s_sleep(1);
                c_io::load(corr_granules, base + lane, raw);
            }
            far = c_io::decode(raw);
        }
        correction[lane] = far;
    }
    if (wave == 0 && block >= 2) {
        granule_word raw[x_io::words];
        const index_type prev2 = base - 2 * B + lane;
        x_io::load(x_granules, prev2, raw);
        while (!__all(x_io::ready(raw)) &&
               !detail::spin_expired(spins, helper)) {
            __builtin_amdgcn_s_sleep(1);
            x_io::load(x_granules, prev2, raw);
        }
        x_prev2_lds[lane] = static_cast<ar_type>(x_io::decode(raw));
    }
    __syncthreads();

    // c = inv(T) rhs_far - M2 x_{b-2} (before the wait for x_{b-1})
    ar_type rhs = own_valid ? own_rhs - correction[lane] : ar_type{};
    {
        ar_type part[2] = {};
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            part[j & 1] += tile[(wave_u * CW + j) * ld + lane] *
                           detail::read_lane(rhs, wave_u * CW + j);
        }
        if (block >= 2) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                part[j & 1] -= m2[j] * x_prev2_lds[wave_u * CW + j];
            }
        }
        partial_b[wave_u * B + lane] = part[0] + part[1];
    }
    if (block > 0) {
        if (wave == 0) {
            granule_word raw[x_io::words];
            const index_type prev = base - B + lane;
            x_io::load(x_granules, prev, raw);
            while (!__all(x_io::ready(raw)) &&
                   !detail::spin_expired(spins, helper)) {
                x_io::load(x_granules, prev, raw);
            }
            x_prev_lds[lane] = static_cast<ar_type>(x_io::decode(raw));
        }
        __syncthreads();
        ar_type part[2] = {};
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            part[j & 1] += panel[(wave_u * CW + j) * ld + lane] *
                           x_prev_lds[wave_u * CW + j];
        }
        partial_a[wave_u * B + lane] = part[0] + part[1];
    }
    __syncthreads();
    if (wave == 0) {
        ar_type sum = (partial_b[lane] + partial_b[B + lane]) +
                      (partial_b[2 * B + lane] + partial_b[3 * B + lane]);
        if (block > 0) {
            sum -= (partial_a[lane] + partial_a[B + lane]) +
                   (partial_a[2 * B + lane] + partial_a[3 * B + lane]);
        }
        const st_type solution = static_cast<st_type>(sum);
        if (own_valid) {
            x_io::publish(x_granules, own_row, solution);
            *own_addr = solution;
        }
    }
}


}  // namespace kernel


/** Hand-off workspace of the split kernel: x granules + correction granules. */
template <typename ArType, typename St>
constexpr std::size_t trsv_split_workspace_bytes(index_type n)
{
    return static_cast<std::size_t>(ceildiv(n, trsv_tile) * trsv_tile) *
           (trsv_granule_words<St>() + trsv_granule_words<ArType>()) *
           sizeof(granule_word);
}

/** Smallest system the split kernel is used for. */
constexpr index_type trsv_split_min_n = 12288;


/**
 * Tries to launch the role-split kernel. Returns hipSuccess and sets
 * `*launched` when the operands and the device allow it; otherwise leaves
 * `*launched` false and launches nothing (the caller falls back to
 * `launch_trsv`). `granules` must hold trsv_split_workspace_bytes().
 */
template <typename ArType, typename MtxView, typename XView>
hipError_t try_launch_trsv_split(bool upper, bool unit_diag,
                                 const MtxView &mtx, const XView &x,
                                 std::uint32_t *helper, granule_word *granules,
                                 int num_cus, hipStream_t stream,
                                 bool *launched)
{
    using mtx_traits = view_traits<MtxView>;
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    constexpr int vec16 = 16 / sizeof(st_type);
    *launched = false;
    const index_type n = mtx.length(0);
    // 16-byte vector path only; 8-byte storage would need 2x the stage
    // registers of the streamers and stays on `kernel::trsv`
    if constexpr (sizeof(st_type) > 4) {
        return hipSuccess;
    } else {
    if (n < trsv_split_min_n) {
        return hipSuccess;
    }
    const bool vec_ok = is_aligned(mtx_traits::data(mtx), 16) &&
                        mtx_traits::stride(mtx) % vec16 == 0 &&
                        (!upper || n % vec16 == 0);
    if (!vec_ok) {
        return hipSuccess;
    }
    constexpr int threads = trsv_split_waves * wave_size;
    const index_type num_blocks = ceildiv(n, trsv_tile);
    const index_type padded = num_blocks * trsv_tile;
    granule_word *x_granules = granules;
    granule_word *corr_granules =
        granules + padded * trsv_granule_words<st_type>();
    const index_type num_words =
        padded * (trsv_granule_words<st_type>() + trsv_granule_words<ArType>());

    hipError_t status = hipSuccess;
#define ACCBLAS_TRSV_SPLIT_LAUNCH(_upper, _unit)                               \
    {                                                                          \
        auto kernel_fn = kernel::trsv_split<_upper, _unit, (vec16 > 1 ? vec16 : 2), ArType, MtxView, \
                                            XView>;                            \
        int per_cu = 0;                                                        \
        status = hipOccupancyMaxActiveBlocksPerMultiprocessor(                 \
            &per_cu, kernel_fn, threads, 0);                                   \
        if (status != hipSuccess) {                                            \
            return status;                                                     \
        }                                                                      \
        /* at most half of the reported capacity for persistent streamers */   \
        const index_type capacity = index_type{per_cu} * num_cus;              \
        index_type streamers = capacity / 2;                                   \
        if (streamers > num_cus) {                                             \
            streamers = num_cus;                                               \
        }                                                                      \
        if (per_cu < 2 || streamers < 1 ||                                     \
            num_blocks > trsv_split_max_owned * streamers) {                   \
            return hipSuccess; /* fall back */                                 \
        }                                                                      \
        const unsigned init_blocks = static_cast<unsigned>(                    \
            std::min<index_type>(ceildiv(num_words, 256), 512));               \
        kernel::trsv_init<<<init_blocks, 256, 0, stream>>>(helper, granules,   \
                                                           num_words);         \
        kernel_fn<<<dim3(static_cast<unsigned>(streamers + num_blocks)),       \
                    dim3(threads), 0, stream>>>(                               \
            mtx, x, helper, x_granules, corr_granules,                         \
            static_cast<int>(streamers));                                      \
        *launched = true;                                                      \
    }
    if (upper) {
        if (unit_diag) {
            ACCBLAS_TRSV_SPLIT_LAUNCH(true, true)
        } else {
            ACCBLAS_TRSV_SPLIT_LAUNCH(true, false)
        }
    } else {
        if (unit_diag) {
            ACCBLAS_TRSV_SPLIT_LAUNCH(false, true)
        } else {
            ACCBLAS_TRSV_SPLIT_LAUNCH(false, false)
        }
    }
#undef ACCBLAS_TRSV_SPLIT_LAUNCH
    return hipGetLastError();
    }  // sizeof(st_type) <= 4
}


}  // namespace dev
}  // namespace accblas
