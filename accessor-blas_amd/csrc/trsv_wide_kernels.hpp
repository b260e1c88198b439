// TRSV with 128-row blocks for gfx950: the kernel for large systems.
//
// Same maths, ticket scheduling, epoch-tagged granule hand-off, one launch per
// solve and accessor contract as `kernel::trsv` (trsv_kernels.hpp; reference:
// `/root/reference/cuda/trsv_kernels.cuh:527-692,725-893`). What changes is
// how much of the dependency chain runs through memory. The solve is a chain
// of hops "x_{b-1} published -> x_b published"; with 64-row blocks every hop
// pays one store -> load round trip through the memory system (0.5 us idle,
// 1-1.5 us beside the solve's own 16 KB-per-block-per-hop load bursts) plus
// ~0.7 us of on-CU work, and rocprofv3 shows the traffic is exactly the
// algorithmic bytes (profiles/r02_trsv_pmc_r01_kernel.json): the time goes
// into the number of round trips, not into wasted bytes. This kernel halves
// them without putting a second sequential step on the CU:
//   * a workgroup of 8 waves owns 128 rows. Its diagonal block
//     T = [D00 0; D10 D11] is inverted completely: the two 64 x 64 tiles by two
//     teams of four waves at the same time (`detail::invert_tile64`), then
//     X10 = -X11 (D10 X00) as two 64 x 64 x 64 products on the matrix cores;
//   * the 128 x 128 panel next to the diagonal is folded with the inverse,
//     W = inv(T) A_{b,b-1} (a 128^3 product on the matrix cores, the panel
//     operand read straight from global memory one pass ahead), so
//     x_b = inv(T) rhs_far - W x_{b-1}: BOTH 64-row halves of x_b follow from
//     x_{b-1} in one step, there is no hop between them;
//   * W is computed TRANSPOSED (W^T = A^T inv(T)^T), which leaves it in the
//     matrix cores' result layout exactly as the mat-vec wants it: wave w
//     holds rows 16w .. 16w+15 of W, a row's 128 entries spread over four
//     lanes of that wave (32 registers each). After x_{b-1} arrives every
//     wave does 32 register-fed FMAs per lane and two cross-lane adds and
//     publishes its own 16 rows: ONE barrier on the critical path (the
//     hand-over of x_{b-1} through LDS), no cross-wave combine;
//   * inv(T) stays in LDS (three 64 x 64 tiles) for c = inv(T) rhs_far, which
//     is computed in the same lane layout before the wait;
//   * far panels: 16 rows per wave, 16-byte loads of the storage type,
//     `Depth` steps in flight per wave, x through LDS (one reader per
//     workgroup, a chunk of published block columns at a time), FMA in the
//     arithmetic type, one transposing cross-lane reduction at the end.
#pragma once

// (matrix loads of the TRSV kernels: non-temporal unless a build says otherwise)
#ifndef ACCBLAS_TRSV_NT
#define ACCBLAS_TRSV_NT true
#endif

#include <hip/hip_runtime.h>

#include "trsv_kernels.hpp"
#include "trsv_wide_detail.hpp"


namespace accblas {
namespace dev {


// Diagnostic builds only (tools/tune_trsv_wide.hip defines
// ACCBLAS_TRSV_WIDE_DIAG): every wave keeps 100 MHz wall-clock stamps of its
// phases in registers and writes them out after it has published, so the
// stamps cost no memory traffic while the block is on the chain. The library
// is never built with it.
#ifdef ACCBLAS_TRSV_WIDE_DIAG
constexpr bool trsv_wide_diag = true;
__device__ unsigned long long *trsv_wide_diag_buffer;
// one record per workgroup, (block * trsv_wide_max_parts + part) * 8 words: start, sweep start,
// first time at the front, block column then, end, first column, end column
__device__ unsigned long long *trsv_wide_diag_workgroups;
#else
constexpr bool trsv_wide_diag = false;
constexpr unsigned long long *trsv_wide_diag_buffer = nullptr;
constexpr unsigned long long *trsv_wide_diag_workgroups = nullptr;
#endif
// (the stamps live in `diag_stamp`, an array of the kernel body that no code is
// generated for unless `trsv_wide_diag`)
#define ACCBLAS_TRSV_WIDE_STAMP(_slot)          \
    do {                                        \
        if constexpr (trsv_wide_diag) {         \
            diag_stamp[_slot] = wall_clock64(); \
        }                                       \
    } while (false)


namespace kernel {


/**
 * One ticket's work of the blocked triangular solve with 128-row blocks (the
 * body of `kernel::trsv_wide`, blockDim = 512). Index mapping (`phys`), granule
 * indexing, tags and the workspace protocol are those of `kernel::trsv`; both
 * kernels can share one workspace. Vec = elements per 16-byte load (any matrix
 * base and row stride, see `trsv_vector_rows`). PanelCols = columns a wave
 * sweeps per step (128, or 64 to halve the registers of a stage), Depth =
 * steps in flight per wave, DeepDepth = the same while the folded panel W is
 * not yet computed (long sweeps defer it). Persistent = the launch has fewer
 * workgroups than tickets and every workgroup draws until none is left.
 * Returns whether the workgroup shall draw again.
 */
namespace detail {
template <bool Upper, bool UnitDiag, int Vec, int PanelCols, int Depth,
          int DeepDepth, typename ArType, bool Persistent, typename MtxView,
          typename XView>
__device__ __forceinline__ bool trsv_wide_ticket(
    index_type unknowns, index_type lda, index_type x_inc,
    const std::remove_cv_t<typename view_traits<XView>::storage_type> *a_data,
    std::remove_cv_t<typename view_traits<XView>::storage_type> *x_data,
    std::uint32_t *helper, trsv_session session, trsv_wide_split split)
{
    // (the views' contents arrive unpacked - `unknowns` = mtx.length(0), row
    // stride, increment, data pointers -: the out-of-line copy of this function
    // then has few enough arguments to get all of them in registers)
    using ar_type = decltype((*static_cast<const MtxView *>(nullptr))(0, 0) *
                             (*static_cast<const XView *>(nullptr))(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    using io = detail::granule_io<st_type>;
    using io_ar = detail::granule_io<ar_type>;  // partial sums of split blocks
    using mma = detail::mma_16x16x4<ar_type>;
    using slot = detail::mma_slot<ar_type>;
    constexpr int R = trsv_wide_rows;    // rows per block
    constexpr int H = trsv_tile;         // 64: tile edge
    constexpr int ld = H + 1;            // LDS leading dimension of a tile
    constexpr int Waves = trsv_wide_waves;
    constexpr int RW = R / Waves;        // 16 rows per wave (sweep and W)
    constexpr int LPR = PanelCols / Vec;  // lanes sharing one panel row
    constexpr int RPI = wave_size / LPR;  // rows per load instruction
    constexpr int NI = RW / RPI;          // load instructions per step
    constexpr int PPB = R / PanelCols;    // sweep steps per 128 columns
    static_assert(Vec > 1 && (PanelCols == 64 || PanelCols == 128) &&
                      LPR <= wave_size && RW % RPI == 0 && Depth >= 1,
                  "bad sweep shape");
    static_assert(RW == 16, "a wave owns one 16-row tile of W");

    // tiles[0] = D00 -> X00, tiles[1] = D11 -> X11 (inverses column-major,
    // tile[c * ld + r]), tiles[2] = D10 (row-major) -> X10 (column-major),
    // tiles[3] = scratch: inversion, Y = D10 X00; later the x chunks
    __shared__ __attribute__((aligned(16))) ar_type tiles[4][H * ld];
    __shared__ ar_type own_lds[R];
    __shared__ ar_type rhs_lds[R];   // own - far correction
    // helpers' sums: all of a block's at once, or `trsv_wide_join_slots` at a
    // time. Nothing of this is on a path that matters - the owner joins once -
    // but the two forms shift the code of the whole kernel: in one process,
    // builds alternated solve by solve (tools/trsv_lib_ab2.py,
    // profiles/r03_trsv_tail.txt), 4-byte storage is 0.3-0.9 % faster with the
    // first form at every size from 8192 to 65536, 2-byte storage 0.6-1.9 %
    // faster with the second.
    constexpr bool join_in_rounds = sizeof(st_type) == 2;
    constexpr int join_rows =
        join_in_rounds ? trsv_wide_join_slots : trsv_wide_max_parts - 1;
    __shared__ ar_type join_lds[join_rows][R];
    __shared__ __attribute__((aligned(16))) st_type x_perm[R];    // x_{b-1} in the order the lanes read it
    __shared__ std::int32_t shared_block;
    __shared__ std::int32_t shared_part;  // part of the block, parts - 1 = owner
    __shared__ std::uint32_t shared_epoch;
    __shared__ bool is_last;

    // Persistent = true: the launch has one workgroup per CU and fewer
    // workgroups than tickets; a workgroup draws tickets until none is left
    // (see `kernel::trsv`: row blocks beyond the CU count would otherwise wait
    // for the hardware dispatcher, which binds workgroup k to XCD k mod 8).
    // Events the last of which re-arms the workspace: every ticket's work done
    // and, when persistent, every workgroup's final draw.
    const unsigned int total_tickets = static_cast<unsigned int>(split.tickets);
    const unsigned int last_event =
        Persistent ? total_tickets + gridDim.x - 1 : total_tickets - 1;
    const int lane = lane_id();
    const int wave_u = __builtin_amdgcn_readfirstlane(wave_id());
    unsigned long long diag_stamp[20] = {};
    unsigned long long diag_poll_issue = 0;   // last probe issued by poll_finish
    unsigned long long diag_poll_probes = 0;  // probes issued by poll_finish
    ACCBLAS_TRSV_WIDE_STAMP(0);  // kernel entry
    const int fi = lane & 15;  // MFMA fragment row (A) / column (B, D)
    const int fk = lane >> 4;  // MFMA fragment k / lane group of a W row
    // `n` = the size the blocks are laid out for (upper: `pad` virtual unknowns
    // in front, see `trsv_wide_virtual_pad`); logical index l exists if
    // pad <= l < n
    const index_type pad =
        trsv_wide_virtual_pad<st_type>(Upper, unknowns);
    const index_type n = unknowns + pad;
    const auto exists = [pad, n](index_type logical) {
        return logical >= pad && logical < n;
    };
    const auto inside = [pad, n](index_type logical) {
        return logical < pad ? pad : (logical < n ? logical : n - 1);
    };
    granule_word *const granules = session.granules;

    if (threadIdx.x == 0) {
        const detail::trsv_ticket drawn =
            detail::draw_ticket(session.state, session.serial, helper,
                                session.granules, session.capacity_words);
        // (a ticket that could not be had - ~0u - maps to a block beyond the
        // matrix: the workgroup leaves, helper[0] says why)
        // (~0u: the wait for a ticket gave up; beyond the last ticket:
        // nothing left to do)
        const unsigned int ticket =
            drawn.ticket < total_tickets
                ? drawn.ticket
                : (drawn.ticket == ~0u ? 0x3fffffffu : 0x3ffffffdu);
        shared_epoch = drawn.epoch;
        // tickets -> (row block, part): blocks below split.first_block have one
        // workgroup; each later block has `parts`, its helpers first and its
        // owner last, so everything a workgroup waits for has a lower ticket
        std::int32_t ticket_block = static_cast<std::int32_t>(ticket);
        std::int32_t ticket_part = 0;
        const unsigned int solvers =
            total_tickets - static_cast<unsigned int>(split.prefetchers);
        if (ticket < total_tickets && ticket >= solvers) {
            // the last tickets do not solve (`trsv_wide_split::prefetchers`)
            ticket_block = detail::trsv_wide_prefetch_block;
            ticket_part = static_cast<std::int32_t>(ticket - solvers);
        }
        if (split.parts > 1 && ticket < total_tickets &&
            ticket_block >= split.first_block &&
            ticket_block != detail::trsv_wide_prefetch_block) {
            const std::int32_t beyond = ticket_block - split.first_block;
            const std::int32_t body =
                (split.tail_block - split.first_block) * split.parts;
            const std::int32_t tail =
                (split.last_block - split.tail_block) * split.tail_parts;
            if (beyond < body) {
                ticket_block = split.first_block + beyond / split.parts;
                ticket_part = beyond % split.parts;
            } else if (beyond - body < tail) {
                ticket_block =
                    split.tail_block + (beyond - body) / split.tail_parts;
                ticket_part = (beyond - body) % split.tail_parts;
            } else {
                ticket_block = split.last_block +
                               (beyond - body - tail) / split.last_parts;
                ticket_part = (beyond - body - tail) % split.last_parts;
            }
        }
        shared_block = ticket_block;
        shared_part = ticket_part;
        if (ticket == 0) {
            // `trsv_init` (cuda/trsv_kernels.cuh:38-42) folded into the first
            // block, as in `kernel::trsv`
            __hip_atomic_store(helper, ~std::uint32_t{0}, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            helper[1] = static_cast<std::uint32_t>(ceildiv(n, index_type{H}));
        }
    }
    __syncthreads();
    ACCBLAS_TRSV_WIDE_STAMP(12);  // ticket drawn
    const index_type block = shared_block;
    const std::uint32_t epoch = shared_epoch;
    // what every workgroup does last (see the end of the kernel)
    const auto finish = [&]() __attribute__((always_inline)) {
        // ---- re-arm the workspace for the next solve (as `kernel::trsv`) -------
        // (`front` was last raised by this block's poller, long before this block
        // counts as finished)
        __syncthreads();
        if (threadIdx.x == 0) {
            drain_vmem();  // this thread's raise of `front` has been performed
            const unsigned int done = __hip_atomic_fetch_add(
                &session.state->finished, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            is_last = done == last_event;
        }
        __syncthreads();
        if (is_last) {
            std::uint32_t next_epoch = epoch + 1;
            if (next_epoch == 0) {
                for (index_type i = threadIdx.x; i < session.capacity_words;
                     i += blockDim.x) {
                    __hip_atomic_store(granules + i, granule_word{0},
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                next_epoch = trsv_first_epoch;
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                detail::rearm_state(session.state, next_epoch);
            }
        }
    };
    if (block == 0x3ffffffd) {
        finish();  // this workgroup's final draw
        return false;
    }
    if (block == 0x3fffffff) {
        return false;  // no ticket to be had: helper[0] says why
    }
    if (block == detail::trsv_wide_prefetch_block) {
        detail::prefetch_ahead<Upper, R>(a_data,
                                         lda, n,
                                         session.state, helper, shared_part,
                                         split.prefetchers);
        finish();
        return Persistent;
    }
    const index_type base = block * R;
    // Split row blocks. A solve with more row blocks than CUs ends with the
    // blocks that have the MOST far columns starting LAST (when a CU frees up),
    // each streaming tens of MB alone while more and more CUs idle: at
    // n = 65536 the second half of the solve ran at 5.1 TB/s against 6.2 TB/s
    // for the first. From `split.first_block` on a block's far columns
    // [0, b - 1 - deferred_columns) are divided evenly over parts - 1 helper
    // workgroups; `parts` consecutive tickets belong to one block: the
    // helpers first - they only sweep their share, published long before they
    // start, and hand their 128 partial sums over as tagged granules - and
    // the owner last: prologue, panel fold, the helpers' sums added in part
    // order (fixed: the result does not depend on timing), the last
    // `deferred_columns` block columns and the tail as usual.
    const int parts_here =
        (split.parts > 1 && block >= split.first_block)
            ? (block >= split.last_block
                   ? split.last_parts
                   : (block >= split.tail_block ? split.tail_parts
                                                : split.parts))
            : 1;
    const int part = shared_part;
    const bool helper_part = part < parts_here - 1;
    granule_word *const partials =
        granules + trsv_wide_partial_offset(n, io::words);
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };

    // this block's right-hand side (never written by another workgroup):
    // loaded ahead of the tiles, used behind them, one wait for all
    st_type own_raw{};
    if (!helper_part) {
        const index_type row = base + threadIdx.x % R;
        own_raw = x_data[phys(inside(row)) * x_inc];
    }

    // ---- stage D00, D11, D10 (a wave reads one tile row per load) ----------
    // All 24 loads of a lane are UNCONDITIONAL (row and column clamped into
    // the matrix; what lies above the diagonal or beyond n is read and
    // dropped) and stay in the storage type until all are back: with the
    // loads inside the conditions the compiler had put a full wait behind each
    // of them, 7.5 us per block (and per solve: block 0 starts the chain).
    if (!helper_part) {
        constexpr int RS = H / Waves;  // tile rows a wave stages
        st_type raw00[RS], raw11[RS], raw10[RS];
        const auto clamped = [&](index_type logical) {
            return phys(inside(logical));
        };
        const index_type col0 = clamped(base + lane);
        const index_type col1 = clamped(base + H + lane);
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int r = wave_u + i * Waves;
            const st_type *row0 = a_data + clamped(base + r) * lda;
            const st_type *row1 = a_data + clamped(base + H + r) * lda;
            raw00[i] = row0[col0];
            raw11[i] = row1[col1];
            raw10[i] = row1[col0];
        }
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int r = wave_u + i * Waves;
            // (gr, gc = row and column of the entry; what does not exist is an
            // identity row / a zero)
            const auto diag_entry = [&](index_type gr, index_type gc,
                                        st_type raw) {
                if (lane == r) {
                    return (UnitDiag || !exists(gr)) ? ar_type{1} : ar_type(raw);
                }
                return (lane < r && exists(gr) && exists(gc)) ? ar_type(raw)
                                                              : ar_type{};
            };
            tiles[0][r * ld + lane] =
                diag_entry(base + r, base + lane, raw00[i]);
            tiles[1][r * ld + lane] =
                diag_entry(base + H + r, base + H + lane, raw11[i]);
            tiles[2][r * ld + lane] =
                exists(base + H + r) && exists(base + lane)
                    ? ar_type(raw10[i])
                    : ar_type{};
        }
    }
    if (!helper_part && threadIdx.x < R) {
        own_lds[threadIdx.x] =
            exists(base + threadIdx.x) ? ar_type(own_raw) : ar_type{};
    }
    __syncthreads();
    ACCBLAS_TRSV_WIDE_STAMP(13);  // tiles staged

    // ---- invert D00 (even waves) and D11 (odd waves) at the same time -------
    // (the serial 16 x 16 substitutions run on team wave 0 = waves 0 and 1,
    // which sit on different SIMDs)
    if (!helper_part) {
        const int team = wave_u % 2;
        ar_type *scratch = tiles[3] + team * 1024;
        detail::invert_tile64<UnitDiag>(tiles[team], scratch, scratch + 512,
                                        wave_u / 2, lane,
                                        (wave_u / 2) * wave_size + lane, 256);
    }
    __syncthreads();
    ACCBLAS_TRSV_WIDE_STAMP(14);  // diagonal tiles inverted

    // ---- X10 = -X11 (D10 X00): 16 output tiles, two per wave ---------------
    if (!helper_part) {
        const int tj = wave_u % 4;
        const int ti_of[2] = {wave_u / 4, 3 - wave_u / 4};
        typename mma::acc_type y[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ti = ti_of[t];
            // X00 is lower triangular: rows above column tile tj are zero
            y[t] = detail::mma_tile_product<ar_type>(
                16 * tj, H, fk,
                [&](int k) { return tiles[2][(16 * ti + fi) * ld + k]; },
                [&](int k) { return tiles[0][(16 * tj + fi) * ld + k]; });
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                tiles[3][(16 * ti_of[t] + mma::row(lane, reg)) * ld + 16 * tj +
                         fi] = y[t][reg];
            }
        }
        __syncthreads();  // Y complete, D10 used up
        typename mma::acc_type z[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ti = ti_of[t];
            // X11 is lower triangular: columns beyond row tile ti are zero
            z[t] = detail::mma_tile_product<ar_type>(
                0, 16 * (ti + 1), fk,
                [&](int k) { return tiles[1][k * ld + 16 * ti + fi]; },
                [&](int k) { return tiles[3][k * ld + 16 * tj + fi]; });
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                tiles[2][(16 * tj + fi) * ld + 16 * ti_of[t] +
                         mma::row(lane, reg)] = -z[t][reg];
            }
        }
    }
    __syncthreads();  // X10 complete: the inverse may be read from here on
    ACCBLAS_TRSV_WIDE_STAMP(15);  // X10 written

    // inv(T)[row][k] for k <= row < 128 (k > row: zero, stored for k in the
    // row's own 64-column tile only)
    const auto inv_entry = [&](int row, int k) -> ar_type {
        if (row < H) {
            return tiles[0][k * ld + row];
        }
        return k < H ? tiles[2][k * ld + row - H]
                     : tiles[1][(k - H) * ld + row - H];
    };
    // Row tile t of W / c costs t + 1 passes over inv(T) (lower triangular).
    // Waves w and w + 4 share a SIMD (and its matrix core, which on gfx950
    // runs fp64 at the vector rate): tiles w and 7 - w give every SIMD 9 passes
    const int own_tile = wave_u < 4 ? wave_u : 11 - wave_u;
    const int own_tile_row = RW * own_tile + fi;  // this lane's row of W / of c
    const int k_end = RW * (own_tile + 1);        // inv(T) row tile stops here

    // ---- W = inv(T) A_{b,b-1}, transposed product, wave w = rows 16w.. -----
    // D[i][j] = sum_k P[k][16 tj + i] inv(T)[16 w + j][k] = W[16w + j][16tj + i]:
    // lane l ends with W[16w + (l & 15)][16 tj + mma::row(l, reg)] in
    // w_acc[tj][reg].
    typename mma::acc_type w_acc[R / 16];
    // Workgroups with a long far sweep ahead of them compute W late (see
    // `deferred_columns`): during the bulk of the sweep its 64 registers hold
    // two more steps of matrix prefetch instead.
    const auto fold_panel = [&]() {
#pragma unroll
        for (int tj = 0; tj < R / 16; ++tj) {
            w_acc[tj] = typename mma::acc_type{};
        }
        // The panel operand comes straight from global memory (the eight
        // waves read the same 64 KB: L1 / L2 hits), one pass of 16 panel rows
        // ahead of the matrix core: no LDS staging, no barriers, every wave
        // runs its own (tile + 1) passes back to back. (Through LDS in four
        // 32-row chunks the fold took 15-23 us for 7.7 us of MFMA time.)
        const index_type prev = base - R;
        using operand = std::remove_cv_t<st_type>;
        const auto fetch = [&](int kp, operand (&a)[R / 16][4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const index_type gr = base + kp + q * 4 + fk;
                const index_type row = phys(inside(gr));
#pragma unroll
                for (int tj = 0; tj < R / 16; ++tj) {
                    // rows that do not exist are masked below
                    a[tj][q] = a_data[row * lda + phys(prev + 16 * tj + fi)];
                }
            }
        };
        operand a_now[R / 16][4];
        operand a_next[R / 16][4];
        fetch(0, a_now);
        for (int kp = 0; kp < k_end; kp += 16) {
            if (kp + 16 < k_end) {
                fetch(kp + 16, a_next);
            }
            ar_type b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b[q] = inv_entry(own_tile_row, kp + q * 4 + fk);
                if (base + kp + q * 4 + fk >= n) {
                    b[q] = ar_type{};  // panel row beyond the matrix
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int tj = 0; tj < R / 16; ++tj) {
                    w_acc[tj] = mma::step(static_cast<ar_type>(a_now[tj][q]),
                                          b[q], w_acc[tj]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int tj = 0; tj < R / 16; ++tj) {
                    a_now[tj][q] = a_next[tj][q];
                }
            }
        }
    };
    // block columns that are swept after W has been computed when W is
    // deferred (W takes ~10 us; the chain's front must still be far away)
    // A row block that is resident from the start (block < split.defer_from)
    // defers nothing: it would follow the front through its bulk columns,
    // fall 7-8 columns behind while it folds (~14 us), and catch up at one tile
    // per memory latency - the pace of the chain itself - so the chain ends up
    // waiting for it (16 us at block 97 of 188, n = 24000, 5 us at every ~10th
    // block behind it: profiles/r04_trsv_wide_front.txt).
    // (a split block always defers: its helpers' sums are joined where it
    // folds)
    const bool defer_fold =
        (block >= split.defer_from ||
         (split.parts > 1 && block >= split.first_block)) &&
        block - 1 >= 2 * trsv_wide_deferred_columns;
    // (the owners of the last blocks start when the front has already passed
    // most of what they would keep, and sweep it with the shallow ring)
    const index_type deferred_columns =
        (split.parts > 1 && block >= split.tail_block)
            ? index_type{split.tail_deferred}
            : trsv_wide_deferred_columns;
    // The first few blocks are due before a fold could finish: they treat
    // x_{b-1} as one more far column (a reduction and the c mat-vec on their
    // hop, 2.4 us instead of 1.7) and start the chain ~15 us earlier. Every
    // block's fold is done 24 us after the launch (slowest wave; the inverse at
    // 12), block 0 publishes at 15 us: with six such blocks (rounds 2-3) the
    // chain stood still from 26.6 to 35 us, when block 6 had folded AND caught
    // up on its far columns; eight: -1...-6 % at n = 4096 ... 8192, -1...-2 % at
    // 24000, eleven and fourteen no better (profiles/r04_trsv_wide_start.txt).
    constexpr index_type unfolded_blocks = 8;
    const bool folded = block >= unfolded_blocks;
    if (folded && !defer_fold && !helper_part) {
        fold_panel();
    }
    if (!folded) {
#pragma unroll
        for (int tj = 0; tj < R / 16; ++tj) {
            w_acc[tj] = typename mma::acc_type{};  // never used
        }
    }

    ACCBLAS_TRSV_WIDE_STAMP(1);  // inverse and W ready
    unsigned int spins = 0;
    // One wave waits for the 128 granules of an earlier block, two per lane
    // (`first + lane` and `first + 64 + lane`): `poll_begin` issues the first
    // probe (so that it can be under way while the wave still has other work),
    // `poll_finish` repeats it until both granules carry this solve's tag.
    const auto poll_begin = [&](index_type first, granule_word (&raw0)[io::words],
                                granule_word (&raw1)[io::words]) {
        io::load(granules, first, raw0);
        io::load(granules, first + H, raw1);
    };
    const auto poll_finish = [&](index_type first, granule_word (&raw0)[io::words],
                                 granule_word (&raw1)[io::words]) {
        if (__all(io::ready(raw0, epoch) && io::ready(raw1, epoch))) {
            return;  // the early probe already found it
        }
        while (!__all(io::ready(raw0, epoch) && io::ready(raw1, epoch)) &&
               !detail::spin_expired(spins, helper)) {
            if constexpr (trsv_wide_diag) {
                diag_poll_issue = wall_clock64();  // issue time of this probe
                ++diag_poll_probes;
            }
            io::load(granules, first, raw0);
            io::load(granules, first + H, raw1);
        }
    };
    // ---- far sweep: correction = sum_{c < 128 (block-1)} A[block][c] x_c ----
    // Step cb covers columns [cb * PanelCols, (cb + 1) * PanelCols); load
    // instruction k of a wave covers rows wave*RW + k*RPI + lane/LPR and Vec
    // consecutive (logical) columns starting at (lane % LPR) * Vec. The
    // matrix data of the next `Depth` steps is always in flight in a register
    // ring (it does not depend on x).
    //
    // x reaches the waves through LDS, a CHUNK of block columns at a time
    // (tiles[3] is free after the prologue): the workgroup meets at a chunk
    // boundary (LDS-only barriers: the matrix prefetches stay in flight), the
    // chunk's x is written to `x_chunk` as stored, and the waves sweep on
    // independently:
    //   * catching up (this workgroup started while the chain's front was
    //     far ahead): `trsv_state::front` tells how many leading blocks are
    //     published; up to `chunk_columns` of them are fetched by all threads
    //     at once. One granule read per element and workgroup instead of one
    //     per element and WAVE, no x registers in the ring (so it can be one
    //     step deeper: the per-CU streaming rate of the late workgroups of a
    //     large solve, 30 GB/s with two steps in flight, was what bounded
    //     n = 65536), and three barriers per `chunk_columns` x 128 columns;
    //   * at the front (fewer than two published columns left): one block
    //     column at a time, polled by wave 0 ALONE on the granules themselves
    //     and handed over through LDS. Eight waves per workgroup polling on
    //     their own made "x_{b-2} seen -> sums reduced -> c", a two-hop path
    //     of the chain, as long as the slowest of eight detections, and ~1500
    //     waves re-reading the newest 1 KB of granules made those cache lines
    //     a hot spot right where the next block's poller and the publisher's
    //     write-through have to get through (profiles/r02_trsv_wide_timeline.txt).
    //     The poller's probe for the next column is issued BEFORE its own
    //     matrix prefetch (loads return in order).
    constexpr int chunk_capacity =
        static_cast<int>(H * ld * sizeof(ar_type) / (R * sizeof(st_type)));
    constexpr int chunk_columns =
        chunk_capacity >= 64 ? 64 : (chunk_capacity >= 32 ? 32 : 16);
    static_assert(chunk_columns <= chunk_capacity, "x chunk must fit tiles[3]");
    st_type *const x_chunk = reinterpret_cast<st_type *>(tiles[3]);
    __shared__ std::int32_t chunk_info[2];  // block columns, x already in LDS
    // block columns this workgroup sweeps: [first_column, far_columns). An
    // unsplit block (or block 0 / 1): [0, b - 1). A split block: its bulk
    // columns [0, b - 1 - deferred_columns) in `parts_here` equal shares, the
    // owner's share running on to b - 1.
    const index_type split_columns =
        parts_here > 1 ? block - 1 - deferred_columns : index_type{0};
    // (the owner takes no share of the bulk columns: it starts no earlier
    // than its helpers, and what it has to do in sequence - prologue, fold,
    // the last `deferred_columns`, the tail - is what the chain ends up
    // waiting for at the end of a large solve)
    const index_type share =
        parts_here > 1 ? ceildiv(split_columns, index_type{parts_here - 1})
                       : index_type{0};
    const index_type first_column =
        parts_here > 1 ? (share * part < split_columns ? share * part
                                                       : split_columns)
                       : index_type{0};
    const index_type far_columns =
        helper_part ? (share * (part + 1) < split_columns ? share * (part + 1)
                                                          : split_columns)
                    : (folded ? block - 1 : block);
    const index_type far_end = far_columns * PPB;    // steps below far_end
    const int row_in_inst = lane / LPR;
    const int col_group = lane % LPR;
    // (a step's matrix data as loaded: the elements are taken apart where they
    // are used. Taken apart where they are loaded - `a[k][e] = v[e]` -, the
    // halves of 2-byte storage cost a shift each right behind the load, in the
    // loop's latch: a full wait per round of the ring)
    struct stage {
        typename packed<st_type, Vec>::type a[NI];
    };
    // (the NI row pointers of a step are formed where they are used, one
    // multiply-add each: as an array they are 16 registers that live through
    // every sweep, and what the compiler spilled to make room for them was
    // reloaded - behind the tile requests in flight - on the block's two-hop
    // path: 0.7 us per hop in the lower instantiation of the front loop,
    // profiles/r04_trsv_wide_front.txt)
    const index_type sweep_row0 = base + wave_u * RW + row_in_inst;
    const auto row_ptr = [&](int k) {
        return a_data + phys(inside(sweep_row0 + k * RPI)) * lda;
    };
    // (upper: first logical column from which whole vectors exist)
    const index_type first_real_vector = ceildiv(pad, index_type{Vec}) * Vec;
    const auto issue = [&](index_type cb, stage &s) {
        const index_type first_col = cb * PanelCols + col_group * Vec;
        // lowest memory column of the group (columns descend for Upper)
        // (upper: vectors of virtual columns only read the first vector of
        // real ones instead - their x is 0 -; the one vector that holds
        // both, when n is not a multiple of Vec, is cleaned up in `consume`)
        const index_type mem_col =
            Upper ? n - 1 -
                        ((first_col + Vec <= pad ? first_real_vector
                                                 : first_col) +
                         Vec - 1)
                  : first_col;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            s.a[k] = load_packed<Vec, ACCBLAS_TRSV_NT>(row_ptr(k) + mem_col);
        }
    };
    ar_type acc[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        acc[k] = ar_type{};
    }
    index_type chunk_first = 0;     // first block column of the chunk in LDS
    index_type chunk_end_step = 0;  // steps below this have their x in LDS
    bool at_front = false;          // caught up: one column at a time from now on
    bool probe_pending = false;     // wave 0: probe of the next column under way
    granule_word probe0[io::words];
    granule_word probe1[io::words];
    // wave 0, at the front: first probe of block column `column`
    const auto begin_probe = [&](index_type column) {
        if (wave_u == 0 && at_front && !probe_pending) {
            poll_begin(column * R + lane, probe0, probe1);
            probe_pending = true;
        }
    };
    // every thread calls this when its next step is `chunk_end_step`
    const auto next_chunk = [&]() {
        const index_type column = chunk_end_step / PPB;
        lds_barrier();  // nobody reads the old chunk any more (the barriers
                        // here must not drain the matrix prefetches)
        if (wave_u == 0) {
            index_type count = 1;
            if (!at_front) {
                // published block columns (relaxed: `front` trails the
                // granules, so what it promises is in memory)
                const std::uint32_t front = __hip_atomic_load(
                    &session.state->front, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
                count = static_cast<index_type>(front) - column;
                count = count < chunk_columns ? count : chunk_columns;
                count = count < far_columns - column ? count
                                                     : far_columns - column;
            }
            if (count < 2) {
                // one block column, by this wave alone (two granules per lane)
                ACCBLAS_TRSV_WIDE_STAMP(10);
                if (!probe_pending) {
                    poll_begin(column * R + lane, probe0, probe1);
                }
                poll_finish(column * R + lane, probe0, probe1);
                probe_pending = false;
                if (lane == 0 && column + 1 == block) {
                    // (an unfolded block meets its predecessor here)
                    __hip_atomic_fetch_max(&session.state->front,
                                           static_cast<unsigned int>(block),
                                           __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                x_chunk[lane] = io::decode(probe0);
                x_chunk[H + lane] = io::decode(probe1);
                ACCBLAS_TRSV_WIDE_STAMP(11);
                count = 1;
            }
            if (lane == 0) {
                chunk_info[0] = static_cast<std::int32_t>(count);
            }
        }
        lds_barrier();
        const int count = chunk_info[0];
        if (count > 1) {
            // all threads fetch the granules of `count` published columns
            constexpr int batch = 8 / io::words;
            const int elements = count * R;
            for (int e0 = 0; e0 < elements; e0 += batch * Waves * wave_size) {
                granule_word raw[batch][io::words];
#pragma unroll
                for (int j = 0; j < batch; ++j) {
                    const int e = e0 + j * Waves * wave_size + threadIdx.x;
                    if (e < elements) {
                        io::load(granules, column * R + e, raw[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < batch; ++j) {
                    const int e = e0 + j * Waves * wave_size + threadIdx.x;
                    if (e < elements) {
                        while (!io::ready(raw[j], epoch) &&
                               !detail::spin_expired(spins, helper)) {
                            io::load(granules, column * R + e, raw[j]);
                        }
                        x_chunk[e] = io::decode(raw[j]);
                    }
                }
            }
            lds_barrier();
        } else {
            if constexpr (trsv_wide_diag) {
                if (diag_stamp[16] == 0) {
                    diag_stamp[16] = wall_clock64();
                    diag_stamp[17] = static_cast<unsigned long long>(column);
                }
            }
            at_front = true;  // from now on the columns come one at a time
        }
        chunk_first = column;
        chunk_end_step = (column + count) * PPB;
    };
    // step cb's matrix data times its x, which lies at `offset` in `x_chunk`
    const auto apply = [&](index_type cb, stage &s, int offset) {
        if (Upper && cb == 0 && pad % Vec != 0) {
            // the vector across the first real column: what was loaded for
            // its virtual columns lies beyond the row's end
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                // (memory element e = logical column Vec-1-e of the group)
                if (col_group * Vec + (Vec - 1 - e) < pad &&
                    col_group * Vec + Vec > pad) {
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        s.a[k][e] = st_type{};
                    }
                }
            }
        }
        ar_type xv[Vec];
#pragma unroll
        for (int e = 0; e < Vec; ++e) {
            xv[e] = static_cast<ar_type>(x_chunk[offset + e]);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                // memory element e of an Upper group is logical column
                // Vec-1-e of the group
                acc[k] += static_cast<ar_type>(s.a[k][e]) *
                          xv[Upper ? Vec - 1 - e : e];
            }
        }
    };
    const auto consume = [&](index_type cb, stage &s) {
        if (cb == chunk_end_step) {
            next_chunk();
        }
        apply(cb, s,
              static_cast<int>(cb / PPB - chunk_first) * R +
                  static_cast<int>(cb % PPB) * PanelCols + col_group * Vec);
        // the poller's probe of the next column goes out before its prefetch
        if (cb + 1 == chunk_end_step && cb + 1 < far_end) {
            begin_probe((cb + 1) / PPB);
        }
    };
    // `value` "depends" on the accumulators: what is computed from it (the
    // addresses of a refill) stays behind the FMAs that produce them
    const auto order_after = [](index_type &value, ar_type (&sums)[NI]) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            asm volatile("" : "+v"(value) : "v"(sums[k]));
        }
    };
    // steps [begin, end) with `depth_tag` steps of matrix data in flight;
    // returns the first step it has not consumed: `end`, or - `until_front_tag`
    // = true - the step behind the one at which the workgroup reached the
    // chain's front (what the ring holds beyond it is dropped: the front loop
    // below requests it again).
    // Whole rounds of the ring whose refills all exist run WITHOUT a condition
    // around the refill: behind `if (cb + D <
    // end)` a slot is a phi of "loaded" and "kept", and the compiler loaded
    // into spare registers and copied them into the slot behind `s_waitcnt
    // vmcnt(7) ... vmcnt(0)` - right behind the step's FMAs, i.e. every step
    // waited for the tile it had just requested and the ring was one step deep
    // in time whatever its depth (the deferred sweep of a late row block, four
    // steps on paper: profiles/r04_trsv_wide_sweep_ring.txt).
    const auto sweep = [&](auto depth_tag, auto until_front_tag,
                           index_type begin, index_type end) -> index_type {
        constexpr int D = decltype(depth_tag)::value;
        constexpr bool until_front = decltype(until_front_tag)::value;
        stage ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (begin + d < end) {
                issue(begin + d, ring[d]);
            }
        }
        index_type cb0 = begin;
        // (the deep ring only - the bulk of a late row block's sweep, before
        // its folded panel exists -: with the shallow sweeps written this way
        // as well the lower instantiations spill 56 bytes and lose 5-12 %)
        // (4-byte storage: n = 36000 ... 49152 -1 ... -6 % depending on the box,
        // 65536 unchanged; 2-byte storage loses 3-5 % with it)
        if (D > Depth && sizeof(st_type) == 4) {
            for (; cb0 + 2 * D <= end; cb0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const index_type cb = cb0 + d;
                    consume(cb, ring[d]);
                    if (until_front && at_front) {
                        return cb + 1;
                    }
                    index_type next = cb + D;
                    order_after(next, acc);
                    issue(next, ring[d]);
                }
            }
        }
        for (; cb0 < end; cb0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const index_type cb = cb0 + d;
                if (cb < end) {
                    consume(cb, ring[d]);
                    if (until_front && at_front) {
                        return cb + 1;
                    }
                    if (cb + D < end) {
                        issue(cb + D, ring[d]);
                    }
                }
            }
        }
        return end;
    };
    // ---- the front loop: one block column per step, the chain's pace --------
    // A workgroup at the front gets one block column of x per hop. Every block
    // is such a follower until its turn, so a follower's turn-around per column
    // bounds the hop. In the general sweep above it is one full latency of the
    // workgroup's own 64 KB tile per column: the poll (vector-memory probes,
    // spin loops with loads) and the refills of the ring share one instruction
    // stream, the compiler cannot count what is outstanding and waits for
    // everything - also for the tile it has just requested - before every step
    // (profiles/r03_trsv_hop_paths.txt, addendum). Here the two jobs have code
    // of their own:
    //   * waves 1..7 (`Poller` = false): a loop with NO vector-memory operation
    //     but the refills of their ring, whole groups of `Depth` steps without
    //     a condition around the refill (it lands in the slot's own registers,
    //     behind the step's FMAs): the wait in front of a step's FMAs counts
    //     the newer refills, the ring is `Depth` tiles deep in time;
    //   * wave 0 (`Poller` = true) polls. Probes return in order behind
    //     whatever the wave has requested before, so it requests its share of
    //     a tile only when it has SEEN the column before - nothing then stands
    //     between two probes - and, so that this late request is served from
    //     the L2 cache and not from HBM, wave 1 has touched those cache lines
    //     one hop earlier (one extra load per step in wave 1's queue).
    // Two LDS-only barriers per column as before. Same arithmetic, same order
    // of summation as the general sweep.
    const auto front_step = [&](auto poller_tag, index_type c, stage &slot,
                                stage &previous, bool refill,
                                std::uint32_t &touched,
                                std::uint32_t &touched_word) {
        constexpr bool Poller = decltype(poller_tag)::value;
        lds_barrier();  // nobody reads the old column's x any more
        if constexpr (Poller) {
            ACCBLAS_TRSV_WIDE_STAMP(10);
            if (!probe_pending) {
                poll_begin(c * R + lane, probe0, probe1);
            }
            poll_finish(c * R + lane, probe0, probe1);
            probe_pending = false;
            ACCBLAS_TRSV_WIDE_STAMP(11);
            if (lane == 0 && c + 1 == block) {
                // (an unfolded block meets its predecessor here)
                __hip_atomic_fetch_max(&session.state->front,
                                       static_cast<unsigned int>(block),
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            x_chunk[lane] = io::decode(probe0);
            x_chunk[H + lane] = io::decode(probe1);
            // this wave's share of the tile `Depth - 1` columns ahead, into
            // the slot of the column consumed one step ago
            // (`refill`: a literal `true` in the main loop below, where the
            // step exists - the request is then unconditional and lands in the
            // slot's own registers. Behind a condition the slot is a phi of
            // "loaded" and "kept": with fp32 arithmetic the compiler loaded
            // into other registers and copied them over behind a full wait -
            // in front of the barrier all eight waves meet at -, and the blocks
            // saw x_{b-2} 1.5-1.8 us after its publication instead of 0.8,
            // profiles/r04_trsv_wide_front.txt section 6)
            // (fp64 arithmetic keeps the condition: its compiler output
            // branches around the loads, no copies, and without the condition
            // the upper instantiations spill - n = 24000: 335 -> 547 us)
            constexpr bool plain_refill = sizeof(ar_type) < 8;
            if (plain_refill ? (refill || c - 1 + Depth < far_end)
                             : (c >= 1 && c - 1 + Depth < far_end)) {
                issue(c - 1 + Depth, previous);
            }
        }
        lds_barrier();
        apply(c, slot, col_group * Vec);
        if constexpr (Poller) {
            if (c + 1 < far_end) {
                poll_begin((c + 1) * R + lane, probe0, probe1);
                probe_pending = true;
            }
        } else {
            if (wave_u == 1 && c + Depth < far_end) {
                // wave 0's rows of the tile it will ask for one hop from now:
                // one 4-byte load per 128-byte line (default cache policy)
                constexpr int line_elements = 128 / static_cast<int>(sizeof(st_type));
                constexpr int lines_per_row = PanelCols / line_elements;
                static_assert(RW * lines_per_row <= wave_size, "one load covers wave 0's rows");
                if (lane < RW * lines_per_row) {
                    const index_type gr = base + lane / lines_per_row;
                    const index_type first_col =
                        (c + Depth) * PanelCols + (lane % lines_per_row) * line_elements;
                    const index_type mem_col =
                        Upper ? n - 1 - (first_col + line_elements - 1) : first_col;
                    // (the value is looked at one step later: looked at here,
                    // the wait for it would be a wait for every refill in flight)
                    touched ^= touched_word;
                    st_type value = a_data[phys(inside(gr)) * lda + (mem_col > 0 ? mem_col : 0)];
                    touched_word = 0;
                    __builtin_memcpy(&touched_word, &value, sizeof(st_type) < 4 ? sizeof(st_type) : 4);
                }
            }
            if (refill) {
                index_type next = c + Depth;
                order_after(next, acc);
                issue(next, slot);
            }
        }
    };
    const auto front_sweep = [&](auto poller_tag, index_type begin,
                                 index_type end) {
        constexpr bool Poller = decltype(poller_tag)::value;
        constexpr int D = Depth;
        stage ring[D];
        std::uint32_t touched = 0;
        std::uint32_t touched_word = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // (the poller's share of step begin + D - 1 is requested by its
            // first step, as of every later one)
            if (begin + d < end &&
                !(Poller && d == D - 1 && (sizeof(ar_type) < 8 || begin >= 1))) {
                issue(begin + d, ring[d]);
            }
        }
        index_type cb0 = begin;
        for (; cb0 + 2 * D <= end; cb0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                front_step(poller_tag, cb0 + d, ring[d], ring[(d + D - 1) % D],
                           true, touched, touched_word);
            }
        }
        for (; cb0 < end; cb0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (cb0 + d < end) {
                    front_step(poller_tag, cb0 + d, ring[d],
                               ring[(d + D - 1) % D], cb0 + d + D < end, touched,
                               touched_word);
                }
            }
        }
        touched ^= touched_word;
        if (touched == 0x5a17c0deu && spins == 0x7fffffffu) {
            helper[1] = touched;  // never: keeps wave 1's touching loads alive
        }
    };
    index_type swept = first_column * PPB;
    chunk_end_step = swept;
    if (helper_part) {
        // a helper's share: bulk columns, nothing else; its 128 sums go to
        // the owner as tagged granules (this path must not run into the
        // owner's code below: the compiler would keep the owner's registers
        // alive across the helper's deep ring)
        sweep(std::integral_constant<int, DeepDepth>{}, std::false_type{}, swept,
              far_end);
        const ar_type total = transpose_reduce_sum<NI, LPR>(acc, lane);
        if (col_group < NI) {
            const int r = wave_u * RW + col_group * RPI + row_in_inst;
            io_ar::publish(partials,
                           (block * (trsv_wide_max_parts - 1) + part) * R + r,
                           total, epoch);
        }
    } else {
    // the helpers' partial sums, added in part order: they are subtracted from
    // this row's right-hand side before the sums of this workgroup's own
    // columns (a fixed order, whatever the timing)
    const auto join_helpers = [&]() __attribute__((always_inline)) {
        // thread t fetches parts t / 128, t / 128 + 4, ... of row t % 128 (one
        // round trip for all of them, a few probes' worth of registers), a
        // thread per row adds them up
        if (parts_here > 1) {
            if constexpr (!join_in_rounds) {
                constexpr int groups = Waves * wave_size / R;
                constexpr int per_thread =
                    (trsv_wide_max_parts - 1 + groups - 1) / groups;
                const int row = threadIdx.x % R;
                const int group = threadIdx.x / R;
                const index_type slot0 =
                    (block * (trsv_wide_max_parts - 1)) * R + row;
                granule_word raw[per_thread][io_ar::words];
#pragma unroll
                for (int j = 0; j < per_thread; ++j) {
                    const int p = group + j * groups;
                    if (p < parts_here - 1) {
                        io_ar::load(partials, slot0 + p * R, raw[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < per_thread; ++j) {
                    const int p = group + j * groups;
                    if (p < parts_here - 1) {
                        while (!io_ar::ready(raw[j], epoch) &&
                               !detail::spin_expired(spins, helper)) {
                            __builtin_amdgcn_s_sleep(4);
                            io_ar::load(partials, slot0 + p * R, raw[j]);
                        }
                        join_lds[p][row] = io_ar::decode(raw[j]);
                    }
                }
                lds_barrier();
                if (threadIdx.x < R) {
                    ar_type helped{};
                    for (int p = 0; p < parts_here - 1; ++p) {
                        helped += join_lds[p][threadIdx.x];
                    }
                    own_lds[threadIdx.x] -= helped;
                }
            } else {
                constexpr int groups = Waves * wave_size / R;
                constexpr int per_thread =
                    (trsv_wide_join_slots + groups - 1) / groups;
                const int row = threadIdx.x % R;
                const int group = threadIdx.x / R;
                const index_type slot0 =
                    (block * (trsv_wide_max_parts - 1)) * R + row;
                ar_type helped{};
                // (`trsv_wide_join_slots` parts at a time through LDS; the sum
                // runs over the parts in order across the rounds)
                for (int p0 = 0; p0 < parts_here - 1; p0 += trsv_wide_join_slots) {
                    granule_word raw[per_thread][io_ar::words];
#pragma unroll
                    for (int j = 0; j < per_thread; ++j) {
                        const int q = group + j * groups;
                        if (q < trsv_wide_join_slots && p0 + q < parts_here - 1) {
                            io_ar::load(partials, slot0 + (p0 + q) * R, raw[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < per_thread; ++j) {
                        const int q = group + j * groups;
                        if (q < trsv_wide_join_slots && p0 + q < parts_here - 1) {
                            while (!io_ar::ready(raw[j], epoch) &&
                                   !detail::spin_expired(spins, helper)) {
                                __builtin_amdgcn_s_sleep(4);
                                io_ar::load(partials, slot0 + (p0 + q) * R, raw[j]);
                            }
                            join_lds[q][row] = io_ar::decode(raw[j]);
                        }
                    }
                    lds_barrier();
                    if (threadIdx.x < R) {
                        for (int q = 0; q < trsv_wide_join_slots &&
                                        p0 + q < parts_here - 1;
                             ++q) {
                            helped += join_lds[q][threadIdx.x];
                        }
                    }
                    lds_barrier();
                }
                if (threadIdx.x < R) {
                    own_lds[threadIdx.x] -= helped;
                }
            }
        }
    };
    if (defer_fold) {
        // the bulk of a long sweep, W not yet computed: a deeper ring. The
        // per-CU streaming rate of the workgroups that start late bounds a
        // large solve (n = 65536: the last block streams 33 MB on its own)
        const index_type bulk_end = (block - 1 - deferred_columns) * PPB;
        if (bulk_end > swept) {
            sweep(std::integral_constant<int, DeepDepth>{}, std::false_type{},
                  swept, bulk_end);
            swept = bulk_end;
        }
        fold_panel();
        join_helpers();
        lds_barrier();
        at_front = false;  // the front moved on meanwhile: catch up in bulk
        chunk_end_step = swept;
        probe_pending = false;
    }
    if (far_end > swept) {
        // 4-byte storage only. Measured (profiles/r04_trsv_wide_front.txt): with
        // the front loop AND without deferred folds an upper solve of 4-byte
        // storage takes 273 instead of 324 us at n = 20000, 333 instead of 390
        // at 24000, 508 instead of 549 at 32768 (equal up to 16384); the lower
        // instantiation the same once its row addresses were formed where
        // they are used (`row_ptr`). 2-byte storage is equal at 8192 and 4-12 %
        // SLOWER from 16384 on with it, whatever its ring's depth
        // (profiles/r05_trsv_f16_front.txt: its general sweep already runs at
        // the hop 4-byte storage needs this loop for).
        if constexpr (sizeof(st_type) == 4 && PPB == 1) {
            // catch up with what is published in chunks, then follow the front
            swept = sweep(std::integral_constant<int, Depth>{}, std::true_type{},
                          swept, far_end);
            if (far_end > swept) {
                if (wave_u == 0) {
                    front_sweep(std::true_type{}, swept, far_end);
                } else {
                    front_sweep(std::false_type{}, swept, far_end);
                }
            }
        } else {
            sweep(std::integral_constant<int, Depth>{}, std::false_type{}, swept,
                  far_end);
        }
    }
    ACCBLAS_TRSV_WIDE_STAMP(2);  // this wave consumed its last far panel
    // this reduction sits between the arrival of x_{b-2} and c, i.e. on the
    // two-hop path of the chain: NI sums per lane over LPR lanes as a
    // transposing butterfly on DPP moves (NI x log2(LPR) LDS-crossbar shuffles
    // took 2 us here, more than a whole hop)
    {
        const ar_type total = transpose_reduce_sum<NI, LPR>(acc, lane);
        if (col_group < NI) {
            const int r = wave_u * RW + col_group * RPI + row_in_inst;
            rhs_lds[r] = exists(base + r) ? own_lds[r] - total : ar_type{};
        }
    }
    ACCBLAS_TRSV_WIDE_STAMP(3);  // this wave's far sums reduced
    lds_barrier();  // (what is handed over lives in LDS)
    ACCBLAS_TRSV_WIDE_STAMP(4);  // every wave's far sums in LDS

    // the poller's first probe for x_{b-1} travels while c is computed
    granule_word tail_raw0[io::words];
    granule_word tail_raw1[io::words];
    if (folded && wave_u == 0) {
        poll_begin(base - R + lane, tail_raw0, tail_raw1);
    }
    // ---- c = inv(T) rhs_far for this lane's row (before the wait) ----------
    // a row's columns are split over its four lanes (k = fk mod 4)
    ar_type c_value;
    {
        ar_type mine{};
        const ar_type *x00 = tiles[0], *x11 = tiles[1], *x10 = tiles[2];
        switch (own_tile) {  // wave-uniform
#define ACCBLAS_TRSV_WIDE_C_CASE(_tile)                                        \
    case _tile:                                                                \
        mine = detail::inverse_row_times<_tile, ld>(x00, x11, x10, rhs_lds,    \
                                                    own_tile_row, fk);         \
        break;
            ACCBLAS_TRSV_WIDE_C_CASE(0)
            ACCBLAS_TRSV_WIDE_C_CASE(1)
            ACCBLAS_TRSV_WIDE_C_CASE(2)
            ACCBLAS_TRSV_WIDE_C_CASE(3)
            ACCBLAS_TRSV_WIDE_C_CASE(4)
            ACCBLAS_TRSV_WIDE_C_CASE(5)
            ACCBLAS_TRSV_WIDE_C_CASE(6)
            ACCBLAS_TRSV_WIDE_C_CASE(7)
#undef ACCBLAS_TRSV_WIDE_C_CASE
        default:
            break;
        }
        c_value = detail::quad_column_sum(mine);
    }

    ACCBLAS_TRSV_WIDE_STAMP(5);  // c ready
    // ---- critical path: x_b = c - W x_{b-1} ---------------------------------
    ar_type solution_value = c_value;
    if (folded) {
        if (wave_u == 0) {
            // ONE wave polls the 128 granules of x_{b-1} (two per lane) and
            // hands them over through LDS as stored (the lanes widen them),
            // already in the lanes' read order: element 16 tj + within goes
            // to lane group quad(within), position 4 tj + reg(within)
            if constexpr (trsv_wide_diag) {
                diag_poll_probes = 0;
                diag_poll_issue = 0;
            }
            poll_finish(base - R + lane, tail_raw0, tail_raw1);
            if constexpr (trsv_wide_diag) {
                diag_stamp[18] = diag_poll_probes;  // probes of the hop's own wait
                diag_stamp[19] = diag_poll_issue;   // issue time of the one that saw it
            }
            if (lane == 0) {
                // blocks 0 .. b-1 are published and visible (just read from
                // memory): tell the workgroups far behind the front
                __hip_atomic_fetch_max(&session.state->front,
                                       static_cast<unsigned int>(block),
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            const int pos =
                slot::quad(lane & 15) * 32 + (lane >> 4) * 4 + slot::reg(lane & 15);
            x_perm[pos] = io::decode(tail_raw0);
            x_perm[pos + 16] = io::decode(tail_raw1);
            ACCBLAS_TRSV_WIDE_STAMP(6);  // x_{b-1} observed (wave 0 only)
        }
        // LDS-only barrier: `__syncthreads()` would make wave 0 wait for its
        // raise of `front` to come back from memory (0.25 us on the hop)
        lds_barrier();
        ACCBLAS_TRSV_WIDE_STAMP(7);  // x_{b-1} handed over
        ar_type part[8] = {};
#pragma unroll
        for (int tj = 0; tj < R / 16; ++tj) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = tj * 4 + reg;
                part[i & 7] +=
                    w_acc[tj][reg] * static_cast<ar_type>(x_perm[fk * 32 + i]);
            }
        }
        const ar_type total = detail::quad_column_sum(
            ((part[0] + part[1]) + (part[2] + part[3])) +
            ((part[4] + part[5]) + (part[6] + part[7])));
        solution_value = c_value - total;
        ACCBLAS_TRSV_WIDE_STAMP(8);  // mat-vec done
    }
    {
        const index_type own_row = base + own_tile_row;
        if (fk == 0 && own_row < n) {
            // (virtual unknowns of an upper solve: handed on - x = 0 -, not
            // stored)
            const st_type solution = static_cast<st_type>(solution_value);
            io::publish(granules, own_row, solution, epoch);  // hand-off first
            if (own_row >= pad) {
                x_data[phys(own_row) * x_inc] = solution;  // the user's vector
            }
        }
    }
    ACCBLAS_TRSV_WIDE_STAMP(9);  // published
    }  // owner
    if constexpr (trsv_wide_diag) {
        if (lane == 0 && trsv_wide_diag_buffer != nullptr && !helper_part) {
            for (int i = 0; i < 20; ++i) {
                trsv_wide_diag_buffer[(block * Waves + wave_u) * 20 + i] =
                    diag_stamp[i];
            }
        }
        if (threadIdx.x == 0 && trsv_wide_diag_workgroups != nullptr) {
            unsigned long long *record =
                trsv_wide_diag_workgroups +
                (block * trsv_wide_max_parts + part) * 8;
            record[0] = diag_stamp[0];
            record[1] = helper_part ? diag_stamp[12] : diag_stamp[1];
            record[2] = diag_stamp[16];
            record[3] = diag_stamp[17];
            record[4] = wall_clock64();
            record[5] = static_cast<unsigned long long>(first_column);
            record[6] = static_cast<unsigned long long>(far_columns);
            record[7] = 1;
        }
    }

    finish();
    return Persistent;
}


/** The same out of line: the row block's registers are allocated apart from
 *  the ticket loop around it. Inlined into `for (;;)`, everything the body
 *  derives from the lane index - a few hundred LDS and matrix addresses - is
 *  invariant in that loop and was computed, and kept, in front of it: 256
 *  registers and 670-780 bytes of scratch, solves 60-80 % slower
 *  (profiles/r04_trsv_persistent.txt). Arguments of a device function travel
 *  in vector registers; what is uniform is brought back into scalar ones, and
 *  the pointers are typed as pointers into global memory (`global_ptr`: as
 *  plain `T *` they were flat pointers here, every matrix load a `flat_load`
 *  that every LDS wait waited for - the first out-of-line build, 6-12 % slower
 *  than one workgroup per ticket, profiles/r05_trsv_persistent.txt). */
template <bool Upper, bool UnitDiag, int Vec, int PanelCols, int Depth,
          int DeepDepth, typename ArType, typename MtxView, typename XView,
          typename StType>
__device__ __noinline__ bool trsv_wide_ticket_call(
    index_type unknowns, index_type lda, index_type x_inc,
    global_ptr<const StType> a_data, global_ptr<StType> x_data,
    global_ptr<std::uint32_t> helper, global_ptr<trsv_state> state,
    global_ptr<granule_word> granules, index_type capacity_words,
    std::uint32_t serial, trsv_wide_split split)
{
    trsv_session session{};
    session.state = (trsv_state *)wave_uniform(state);
    session.granules = (granule_word *)wave_uniform(granules);
    session.capacity_words = wave_uniform(capacity_words);
    session.serial = wave_uniform(serial);
    return trsv_wide_ticket<Upper, UnitDiag, Vec, PanelCols, Depth, DeepDepth,
                            ArType, true, MtxView, XView>(
        wave_uniform(unknowns), wave_uniform(lda), wave_uniform(x_inc),
        (const StType *)wave_uniform(a_data), (StType *)wave_uniform(x_data),
        (std::uint32_t *)wave_uniform(helper), session, wave_uniform(split));
}
}  // namespace detail


/**
 * Blocked triangular solve with 128-row blocks. blockDim = 512, grid = the
 * solve's tickets (`trsv_wide_tickets`), or - Persistent - one workgroup per CU
 * drawing tickets until none is left: row blocks beyond the CU count would
 * otherwise wait for the hardware dispatcher, which binds workgroup k of a
 * launch to XCD k mod 8 (see `kernel::trsv`).
 */
template <bool Upper, bool UnitDiag, int Vec, int PanelCols, int Depth,
          int DeepDepth, typename ArType, bool Persistent = false,
          typename MtxView, typename XView>
__global__ __launch_bounds__(trsv_wide_waves *wave_size) void trsv_wide(
    MtxView mtx, XView x, std::uint32_t *helper, trsv_session session,
    trsv_wide_split split)
{
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    const st_type *a_data = view_traits<MtxView>::data(mtx);
    st_type *x_data = view_traits<XView>::data(x);
    const index_type unknowns = mtx.length(0);
    const index_type lda = view_traits<MtxView>::stride(mtx);
    const index_type x_inc = view_traits<XView>::stride(x);
    if constexpr (Persistent) {
        while (detail::trsv_wide_ticket_call<Upper, UnitDiag, Vec, PanelCols,
                                             Depth, DeepDepth, ArType, MtxView,
                                             XView, st_type>(
            unknowns, lda, x_inc, (global_ptr<const st_type>)a_data,
            (global_ptr<st_type>)x_data, (global_ptr<std::uint32_t>)helper,
            (global_ptr<trsv_state>)session.state,
            (global_ptr<granule_word>)session.granules, session.capacity_words,
            session.serial, split)) {
            __syncthreads();  // the next ticket's work re-uses what lives in LDS
        }
    } else {
        detail::trsv_wide_ticket<Upper, UnitDiag, Vec, PanelCols, Depth,
                                 DeepDepth, ArType, false, MtxView, XView>(
            unknowns, lda, x_inc, a_data, x_data, helper, session, split);
    }
}


}  // namespace kernel


}  // namespace dev
}  // namespace accblas


#include "trsv_wide_launch.hpp"
