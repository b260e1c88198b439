// TRSV with 128-row blocks for gfx950 (large, 16-byte-aligned systems).
//
// Same maths, ticket scheduling, granule hand-off and accessor contract as
// `kernel::trsv` (trsv_kernels.hpp; reference:
// `/root/reference/cuda/trsv_kernels.cuh:527-692,725-893`). The solve is a
// chain of dependent hops through memory (>= 0.5 us each on MI355X, ~1.2-2 us
// with the work that sits on the path). `kernel::trsv` needs n/64 of them;
// this kernel needs n/128, with the same amount of work on the path per hop:
//   * a workgroup of 8 waves owns 128 rows; its diagonal tile
//     T = [D00 0; D10 D11] is inverted blockwise (two 64 x 64 inversions by two
//     waves at the same time, then X10 = -inv(D11) D10 inv(D00) by everybody);
//   * the 128 x 128 panel next to the diagonal is folded with the inverse,
//     W = inv(T) A_{b,b-1}, so x_b = inv(T) rhs_far - W x_{b-1};
//   * W lives in REGISTERS, distributed over the workgroup: thread (half h,
//     column group g, lane l) holds 32 consecutive entries of row 64 h + l
//     (64 VGPRs for fp64). The 128 x 128 mat-vec on the critical path is then
//     32 register-fed FMAs per thread (vector entries broadcast from LDS) plus
//     one LDS combine over the 4 column groups; inv(T) stays in LDS as three
//     64 x 64 tiles (a 128 x 128 fp64 tile alone would be 128 KiB);
//   * far panels (128 columns wide): 16 rows per wave, 16-byte loads, double
//     buffered with the x granules, one cross-lane reduction at the end.
#pragma once

#include <hip/hip_runtime.h>

#include "trsv_kernels.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_wide_tile = 128;
constexpr int trsv_wide_waves = 8;
constexpr index_type trsv_wide_min_n = 4096;


namespace kernel {


template <bool Upper, bool UnitDiag, int Vec, typename ArType,
          typename MtxView, typename XView>
__global__ __launch_bounds__(trsv_wide_waves *wave_size) void trsv_wide(
    MtxView mtx, XView x, std::uint32_t *helper, granule_word *granules)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    using io = detail::granule_io<st_type>;
    constexpr int T = trsv_wide_tile;   // rows per block
    constexpr int H = T / 2;            // 64: half tile = wave width
    constexpr int ld = H + 1;           // LDS leading dimension of 64x64 tiles
    constexpr int Waves = trsv_wide_waves;
    constexpr int G = Waves / 2;        // column groups per half (4)
    constexpr int CG = T / G;           // columns per group (32)
    constexpr int RW = T / Waves;       // rows per wave in the far sweep (16)
    constexpr int LPR = T / Vec;        // lanes sharing one panel row
    constexpr int RPI = wave_size / LPR;  // rows per load instruction
    constexpr int NI = RW / RPI;        // load instructions per panel
    static_assert(Vec > 1 && LPR <= wave_size && RW % RPI == 0, "bad Vec");

    // four 64x64 staging tiles; the last one doubles as the panel chunk buffer
    __shared__ ar_type buf[4][H * ld];
    __shared__ ar_type correction[T];
    __shared__ ar_type own_lds[T];
    __shared__ ar_type x_prev_lds[T];
    __shared__ ar_type part[Waves * H];
    __shared__ std::int32_t shared_block;

    const int lane = lane_id();
    const int wave = wave_id();
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int half = wave_u / G;      // 0: rows 0..63, 1: rows 64..127
    const int group = wave_u % G;     // column group of this wave
    const int row = half * H + lane;  // tile-local row of this thread
    const index_type n = mtx.length(0);

    if (threadIdx.x == 0) {
        shared_block = static_cast<std::int32_t>(
            __hip_atomic_fetch_add(helper + 1, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    const index_type block = shared_block;
    const index_type base = block * T;
    if (base >= n) {
        return;
    }
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };
    const st_type *a_data = mtx_traits::data(mtx);
    const index_type lda = mtx_traits::stride(mtx);
    st_type *x_data = x_traits::data(x);
    const index_type x_inc = x_traits::stride(x);
    // element (logical row gr, logical column gc) of the matrix, 0 if the row
    // does not exist
    const auto entry = [&](index_type gr, index_type gc) {
        return gr < n ? ar_type(mtx(phys(gr), phys(gc))) : ar_type{};
    };

    const index_type own_row = base + row;
    const bool own_valid = own_row < n;
    st_type *own_addr = x_data + phys(own_valid ? own_row : n - 1) * x_inc;
    ar_type own_rhs{};
    if (own_valid) {
        own_rhs = static_cast<ar_type>(*own_addr);
    }
    if (group == 0) {
        own_lds[row] = own_rhs;  // read after the barriers below
    }

    // ---- stage D00 -> buf[0], D11 -> buf[1] (row-major, diagonal tiles) and
    //      D10 -> buf[2] (row-major) ------------------------------------------
    for (int r = wave; r < H; r += Waves) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const index_type gr = base + d * H + r;
            const index_type gc = base + d * H + lane;
            ar_type value{};
            if (lane == r) {
                value = (UnitDiag || gr >= n) ? ar_type{1} : entry(gr, gc);
            } else if (lane < r) {
                value = entry(gr, gc);
            }
            buf[d][r * ld + lane] = value;
        }
        buf[2][r * ld + lane] = entry(base + H + r, base + lane);
    }
    __syncthreads();

    // ---- invert the two diagonal tiles (waves 0 and G), lane = column -------
    if (wave_u == 0 || wave_u == G) {
        ar_type *tile = buf[wave_u == 0 ? 0 : 1];
        ar_type column[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
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
        for (int i = 0; i < H; ++i) {
            tile[lane * ld + i] = column[i];  // inverse, column-major
        }
    }
    __syncthreads();

    // ---- X10 = -inv(D11) * D10 * inv(D00), 8 output columns per wave -------
    // inv(D00)[k][j] = buf[0][j*ld + k], inv(D11)[i][k] = buf[1][k*ld + i]
    {
        constexpr int OC = H / Waves;  // 8
        const int j0 = wave_u * OC;
        ar_type y[OC];
#pragma unroll
        for (int q = 0; q < OC; ++q) {
            y[q] = ar_type{};
        }
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const ar_type d = buf[2][lane * ld + k];  // D10[i = lane][k]
#pragma unroll
            for (int q = 0; q < OC; ++q) {
                y[q] += d * buf[0][(j0 + q) * ld + k];
            }
        }
#pragma unroll
        for (int q = 0; q < OC; ++q) {
            buf[3][(j0 + q) * ld + lane] = y[q];  // Y = D10 inv(D00), col-major
        }
        __syncthreads();
        ar_type z[OC];
#pragma unroll
        for (int q = 0; q < OC; ++q) {
            z[q] = ar_type{};
        }
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const ar_type d = buf[1][k * ld + lane];  // inv(D11)[i = lane][k]
#pragma unroll
            for (int q = 0; q < OC; ++q) {
                z[q] -= d * buf[3][(j0 + q) * ld + k];
            }
        }
#pragma unroll
        for (int q = 0; q < OC; ++q) {
            buf[2][(j0 + q) * ld + lane] = z[q];  // X10, col-major (D10 is dead)
        }
    }
    __syncthreads();

    // inv(T)[r][k] for this thread's row r (tile-local) and any column k
    const auto inv_entry = [&](int k) -> ar_type {
        if (half == 0) {
            return k < H ? buf[0][k * ld + lane] : ar_type{};
        }
        return k < H ? buf[2][k * ld + lane] : buf[1][(k - H) * ld + lane];
    };

    // ---- W = inv(T) * A_{b,b-1}: this thread's 32 entries --------------------
    ar_type w_reg[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) {
        w_reg[q] = ar_type{};
    }
    if (block > 0) {
        constexpr int KC = 32;  // panel rows per chunk: 32 x 128 values in buf[3]
        ar_type *chunk = buf[3];
        for (int k0 = 0; k0 < T; k0 += KC) {
            __syncthreads();  // previous chunk (or Y) fully consumed
            // 4096 values / 512 threads: rows wave*4 .. wave*4+3, two columns
#pragma unroll
            for (int j = 0; j < KC / Waves; ++j) {
                const int kl = wave_u * (KC / Waves) + j;
                const index_type gr = base + k0 + kl;
                chunk[kl * T + lane] = entry(gr, base - T + lane);
                chunk[kl * T + H + lane] = entry(gr, base - T + H + lane);
            }
            __syncthreads();
            if (half == 1 || k0 < H) {  // inv(T)[r][k] = 0 for r < 64 <= k
                for (int kl = 0; kl < KC; ++kl) {
                    const ar_type t = inv_entry(k0 + kl);
#pragma unroll
                    for (int q = 0; q < CG; ++q) {
                        w_reg[q] += t * chunk[kl * T + group * CG + q];
                    }
                }
            }
        }
    }

    // ---- far sweep: correction = sum_{c < block-1} A[block][c] x_c ---------
    const index_type far_end = block - 1;  // panels [0, far_end), 128 columns
    const int row_in_inst = lane / LPR;
    const int col_group = lane % LPR;
    struct stage {
        st_type a[NI][Vec];
        granule_word xraw[Vec][io::words];
    };
    const st_type *row_ptr[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const index_type gr = base + wave_u * RW + k * RPI + row_in_inst;
        row_ptr[k] = a_data + phys(gr < n ? gr : n - 1) * lda;
    }
    const auto issue = [&](index_type cb, stage &s) {
        const index_type first_col = cb * T + col_group * Vec;  // logical
        const index_type mem_col =
            Upper ? n - 1 - (first_col + Vec - 1) : first_col;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const auto v = load_packed<Vec, true>(row_ptr[k] + mem_col);
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                s.a[k][e] = v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < Vec; ++e) {
            io::load(granules, first_col + e, s.xraw[e]);
        }
    };
    ar_type acc[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        acc[k] = ar_type{};
    }
    unsigned int spins = 0;
    const auto consume = [&](index_type cb, stage &s) {
        const index_type first_col = cb * T + col_group * Vec;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                ok = ok && io::ready(s.xraw[e]);
            }
            if (__all(ok) || detail::spin_expired(spins, helper)) {
                break;
            }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                io::load(granules, first_col + e, s.xraw[e]);
            }
        }
        ar_type xv[Vec];
#pragma unroll
        for (int e = 0; e < Vec; ++e) {
            xv[e] = static_cast<ar_type>(io::decode(s.xraw[e]));
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                acc[k] += static_cast<ar_type>(s.a[k][e]) *
                          xv[Upper ? Vec - 1 - e : e];
            }
        }
    };
    if (far_end > 0) {
        stage even, odd;
        issue(0, even);
        index_type cb = 0;
        for (; cb + 1 < far_end; cb += 2) {
            issue(cb + 1, odd);
            consume(cb, even);
            if (cb + 2 < far_end) {
                issue(cb + 2, even);
            }
            consume(cb + 1, odd);
        }
        if (cb < far_end) {
            consume(cb, even);
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const ar_type total = group_all_reduce<LPR>(
            acc[k], [](ar_type a, ar_type b) { return a + b; });
        if (col_group == 0) {
            const int r = wave_u * RW + k * RPI + row_in_inst;
            correction[r] = (base + r < n) ? total : ar_type{};
        }
    }
    __syncthreads();

    // ---- tail: x_b = inv(T) rhs_far - W x_{b-1} -----------------------------
    // c = inv(T) rhs_far with rhs_far[k] = own[k] - correction[k]: both vectors
    // are in LDS for all 128 rows, inv(T) is read from the staging tiles
    // (this mat-vec is done before the wait; only W needs to be in registers)
    ar_type sum[4] = {};
#pragma unroll
    for (int q = 0; q < CG; ++q) {
        const int k = group * CG + q;
        const ar_type rhs_k =
            (base + k < n) ? own_lds[k] - correction[k] : ar_type{};
        sum[q & 3] += inv_entry(k) * rhs_k;
    }
    if (block > 0) {
        // critical path: waves 0 and G poll the two halves of x_{b-1}
        if (group == 0) {
            granule_word raw[io::words];
            const index_type prev = base - T + half * H + lane;  // always < n
            io::load(granules, prev, raw);
            while (!__all(io::ready(raw)) &&
                   !detail::spin_expired(spins, helper)) {
                io::load(granules, prev, raw);
            }
            x_prev_lds[half * H + lane] =
                static_cast<ar_type>(io::decode(raw));
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CG; ++q) {
            sum[q & 3] -= w_reg[q] * x_prev_lds[group * CG + q];
        }
    }
    part[wave_u * H + lane] = (sum[0] + sum[1]) + (sum[2] + sum[3]);
    __syncthreads();
    if (group == 0) {
        const ar_type *mine = part + half * G * H;
        const ar_type total = (mine[lane] + mine[H + lane]) +
                              (mine[2 * H + lane] + mine[3 * H + lane]);
        const st_type solution = static_cast<st_type>(total);
        if (own_valid) {
            io::publish(granules, own_row, solution);  // hand-off first
            *own_addr = solution;                       // the user's vector
        }
    }
}


}  // namespace kernel


/**
 * Launches the 128-row kernel when the operands allow it (16-byte aligned
 * matrix base and row stride, Upper: n % Vec == 0, storage <= 4 bytes,
 * n >= trsv_wide_min_n); `*launched` tells. The workspace is the same as for
 * `launch_trsv` (one granule entry per element of x).
 */
template <typename ArType, typename MtxView, typename XView>
hipError_t try_launch_trsv_wide(bool upper, bool unit_diag, const MtxView &mtx,
                                const XView &x, std::uint32_t *helper,
                                granule_word *granules, hipStream_t stream,
                                bool *launched)
{
    using mtx_traits = view_traits<MtxView>;
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    *launched = false;
    if constexpr (sizeof(st_type) > 4) {
        return hipSuccess;
    } else {
        constexpr int vec16 = 16 / sizeof(st_type);
        const index_type n = mtx.length(0);
        const bool vec_ok = is_aligned(mtx_traits::data(mtx), 16) &&
                            mtx_traits::stride(mtx) % vec16 == 0 &&
                            (!upper || n % vec16 == 0);
        if (n < trsv_wide_min_n || !vec_ok) {
            return hipSuccess;
        }
        const dim3 grid(static_cast<unsigned>(ceildiv(n, trsv_wide_tile)));
        const dim3 block(trsv_wide_waves * wave_size);
        const index_type num_words = n * trsv_granule_words<st_type>();
        const unsigned init_blocks = static_cast<unsigned>(
            std::min<index_type>(ceildiv(num_words, 256), 512));
        kernel::trsv_init<<<init_blocks, 256, 0, stream>>>(helper, granules,
                                                           num_words);
        if (upper) {
            if (unit_diag) {
                kernel::trsv_wide<true, true, vec16, ArType>
                    <<<grid, block, 0, stream>>>(mtx, x, helper, granules);
            } else {
                kernel::trsv_wide<true, false, vec16, ArType>
                    <<<grid, block, 0, stream>>>(mtx, x, helper, granules);
            }
        } else {
            if (unit_diag) {
                kernel::trsv_wide<false, true, vec16, ArType>
                    <<<grid, block, 0, stream>>>(mtx, x, helper, granules);
            } else {
                kernel::trsv_wide<false, false, vec16, ArType>
                    <<<grid, block, 0, stream>>>(mtx, x, helper, granules);
            }
        }
        *launched = true;
        return hipGetLastError();
    }
}


}  // namespace dev
}  // namespace accblas
