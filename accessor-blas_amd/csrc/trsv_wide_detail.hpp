// Pieces of the 128-row TRSV kernel (`kernel::trsv_wide`,
// trsv_wide_kernels.hpp) that stand by themselves: the shape constants and the
// split of a solve into tickets, the prefetching workgroups' loop, the matrix
// cores' result layout, and c = inv(T) rhs for one row.
#pragma once

#include <hip/hip_runtime.h>

#include "trsv_kernels.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_wide_rows = 128;
constexpr int trsv_wide_waves = 8;

/** Block columns next to its diagonal that a workgroup with a long far sweep
 *  keeps for after the panel fold (see `defer_fold` in the kernel). */
constexpr index_type trsv_wide_deferred_columns = 48;

/** The virtual unknowns in front of an upper solve (`trsv_virtual_pad`). */
template <typename StType>
constexpr index_type trsv_wide_virtual_pad(bool upper, index_type n)
{
    return trsv_virtual_pad<StType>(upper, n);
}

/** Row blocks from `first_block` on are worked on by `parts` workgroups each
 *  (see `kernel::trsv_wide`); parts == 1 switches the splitting off. */
struct trsv_wide_split {
    int first_block;
    int parts;
    int tail_block;  // blocks from here on have `tail_parts` workgroups each
    int tail_parts;  // (smaller shares: the last tickets run almost alone)
    int tail_deferred;  // block columns the owner of a tail block keeps
    int last_block;  // blocks from here on (>= tail_block): `last_parts` each,
    int last_parts;  // their owners keep `tail_deferred` columns as well
    // Workgroups beyond the ones that solve (the last tickets): they run on
    // CUs the solve leaves idle and pull the tiles of the next block columns
    // into the Infinity Cache ahead of the chain's front (see
    // `detail::prefetch_ahead`)
    int prefetchers;
    // Tickets of the solve (solving workgroups + prefetchers): what
    // `trsv_wide_grid` returns. A launch may have fewer workgroups than
    // tickets (one per CU): they draw tickets until none is left.
    int tickets;
    // Row blocks from this one on fold their panel late when their far sweep
    // is long (see `defer_fold` in the kernel): the row blocks that START late.
    // The ones that have a CU from the first moment of the solve (set by the
    // launcher: all of them, or the first `num_cus`) fold at once, while the
    // chain is still far away.
    int defer_from;
};
constexpr int trsv_wide_max_parts = trsv_partial_slots + 1;
// helpers' sums an owner takes through LDS at a time
constexpr int trsv_wide_join_slots = 7;

/** Granule words of the partial-sum area that follows the x granules in the
 *  workspace: (parts - 1) x 128 partial sums per row block, one or two words
 *  each (arithmetic type), tagged like the x granules. Sized for fp64
 *  arithmetic and the maximum number of parts. */
constexpr index_type trsv_wide_partial_words(index_type n)
{
    return ceildiv(n, index_type{trsv_wide_rows}) * (trsv_wide_max_parts - 1) *
           trsv_wide_rows * 2;
}

/** First word of that area (64-word aligned behind n x words_per_element). */
constexpr index_type trsv_wide_partial_offset(index_type n, int words_per_element)
{
    return ceildiv(n * words_per_element, index_type{64}) * 64;
}



namespace kernel {
namespace detail {


/** Block columns a prefetching workgroup keeps ahead of the chain's front. */
constexpr int trsv_wide_prefetch_ahead = 5;
/** What a prefetching workgroup finds in place of a row block. */
constexpr std::int32_t trsv_wide_prefetch_block = 0x3ffffffe;

/**
 * The life of a prefetching workgroup (`trsv_wide_split::prefetchers`): it
 * follows `trsv_state::front` and touches every cache line of the tiles
 * (row block b, block column c) for c up to `trsv_wide_prefetch_ahead` columns
 * beyond the front and b = c + 1 + index, c + 1 + index + count, ... - one
 * 4-byte load per 128-byte line and thread, the values are thrown away. Why:
 * a follower's turn-around per block column is the latency of its own tile
 * (the compiler's loop over the register ring waits for the tile it has just
 * issued, see profiles/r03_trsv_hop_paths.txt), every block is a follower
 * until its turn, and a tile that waits in the Infinity Cache arrives sooner
 * than one that comes from HBM. Measured as a kernel of its own beside the
 * solve (tools/tune_trsv_prefetch.hip, profiles/r03_trsv_prefetch.txt, 4-byte
 * storage): n = 6144 -6 %, 8192 -5 %, 12288 -6 %, 16384 -4.5 %, 18432 -2.5 %,
 * 20480 +1.5 %, 24000 +4.7 % (there the second copy of 12 MB per hop is more
 * than the fabric has to spare); as workgroups of the solve itself (library
 * builds alternated in one process): 4096 -3.9 %, 6144 -6.2 %, 12288 -5 %,
 * 16384 -2.5 %; 8-byte storage 4096 -15 %, 8192 -26 %, 16384 -12 % (its
 * tiles are twice as large); 2-byte storage +5 % (not used there). Returns when the front is within three block
 * columns of the end, when a solver workgroup has given up, or after
 * `trsv_spin_limit` looks at a front that has not moved meanwhile; it never
 * writes the error word itself.
 */
template <bool Upper, int R, typename StType>
__device__ __forceinline__ void prefetch_ahead(const StType *a_data,
                                               index_type lda, index_type n,
                                               const trsv_state *state,
                                               std::uint32_t *helper,
                                               int index, int count)
{
    constexpr int line_elements = 128 / static_cast<int>(sizeof(StType));
    constexpr int lines_per_row = R / line_elements;  // of one tile row
    constexpr int lines = R * lines_per_row;          // of one tile
    const int blocks = static_cast<int>(ceildiv(n, index_type{R}));
    int done = 2;  // block columns below this are taken care of
    unsigned int spins = 0;
    int last_front = -1;
    std::uint32_t bits = 0;
    while (true) {
        const int front = static_cast<int>(
            __hip_atomic_load(&state->front, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT));
        if (front + 3 >= blocks) {
            break;
        }
        const int upto = front + 1 + trsv_wide_prefetch_ahead < blocks - 1
                             ? front + 1 + trsv_wide_prefetch_ahead
                             : blocks - 1;
        done = done < front + 1 ? front + 1 : done;  // (what the front has passed)
        if (front != last_front) {
            last_front = front;
            spins = 0;  // the limit is on a front that does not MOVE
        }
        if (done >= upto) {
            // A prefetcher is optional: it leaves when a solver has given up
            // or when the front has stood still for `trsv_spin_limit` looks,
            // but it never declares the solve failed (a solve stretched by
            // other streams' kernels completes without it).
            ++spins;
            if (spins >= trsv_spin_limit ||
                ((spins & 1023u) == 0 &&
                 __hip_atomic_load(helper, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) ==
                     trsv_error_word)) {
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        for (int c = done; c < upto; ++c) {
            for (int b = c + 1 + index; b < blocks; b += count) {
                for (int line = static_cast<int>(threadIdx.x); line < lines;
                     line += static_cast<int>(blockDim.x)) {
                    const index_type row =
                        index_type{b} * R + line / lines_per_row;
                    const index_type col = index_type{c} * R +
                                           (line % lines_per_row) * line_elements;
                    if (row < n) {
                        const index_type mem_row = Upper ? n - 1 - row : row;
                        const index_type mem_col =
                            Upper ? n - 1 - (col + line_elements - 1) : col;
                        StType value = a_data[mem_row * lda +
                                              (mem_col > 0 ? mem_col : 0)];
                        std::uint32_t word = 0;
                        __builtin_memcpy(&word, &value,
                                         sizeof(StType) < 4 ? sizeof(StType) : 4);
                        bits ^= word;
                    }
                }
            }
        }
        done = upto;
    }
    if (bits == 0x5a17c0deu && spins == 0x7fffffffu) {
        helper[1] = bits;  // never: keeps the loads alive
    }
}


/** Where element `within` (0..15) of a 16-wide row of an MFMA result tile
 *  lives: lane group `quad` (lane >> 4) and accumulator register `reg`, the
 *  inverse of `mma_16x16x4<T>::row`. */
template <typename T>
struct mma_slot;

template <>
struct mma_slot<double> {
    __device__ __forceinline__ static int quad(int within) { return within & 3; }
    __device__ __forceinline__ static int reg(int within) { return within >> 2; }
};

template <>
struct mma_slot<float> {
    __device__ __forceinline__ static int quad(int within) { return within >> 2; }
    __device__ __forceinline__ static int reg(int within) { return within & 3; }
};


/** Sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (all four end with it). */
template <typename T>
__device__ __forceinline__ T quad_column_sum(T value)
{
    // (value + the lane 16 away, then + the lane 32 away, as two
    // `__shfl_xor` steps would give - without their trips through LDS: this
    // sum ends both c and W x on the chain's hop)
    value = lane_bit_pair_sum<4>(value);
    return lane_bit_pair_sum<5>(value);
}


/**
 * This lane's share of c = inv(T) rhs for a row of row tile `Tile` (rows
 * 16 Tile .. 16 Tile + 15 of the 128-row block; `row` = this lane's row inside
 * the block). The row's columns k < 16 (Tile + 1) are split over its four
 * lanes in runs of eight: lane group fk takes k = 32 m + 8 fk + j, j < 8.
 *   * Eight columns apart, the four lanes of a row - and with them the 64
 *     lanes of the wave, whose rows are consecutive - fall on bank pairs
 *     (8 fk + fi) mod 32, every pair exactly twice: the minimum for 64 8-byte
 *     reads. With neighbouring columns per lane (k = 16 s + 4 q + fk, the
 *     layout until round 3) the pairs were (fk + fi) mod 32, up to four lanes
 *     each, and the eight waves' reads queued in the one LDS: 0.64 us between
 *     "all far sums in LDS" and "c ready" for the slowest wave, on the chain's
 *     two-hop path x_{b-2} -> c -> x_b (profiles/r03_trsv_hop_paths.txt).
 *   * The trip count is a compile-time constant: every LDS read of the row is
 *     issued before the first FMA needs one.
 * `x00`, `x11`, `x10` = the inverse's three 64 x 64 tiles, column-major with
 * leading dimension `ld`. Four partial sums per lane (j mod 4), ascending k.
 */
template <int Tile, int ld, typename T>
__device__ __forceinline__ T inverse_row_times(const T *x00, const T *x11,
                                               const T *x10, const T *rhs,
                                               int row, int fk)
{
    constexpr int H = 64;
    constexpr int columns = 16 * (Tile + 1);
    constexpr int runs = (columns + 31) / 32;
    // (two runs - 16 entries and 16 right-hand sides, 64 registers with fp64 -
    // are read ahead of their FMAs at a time: all four at once made this the
    // place where the kernel needs the most registers, and what is live across
    // it - W, the row pointers of the sweep - paid for that with spills)
    constexpr int batch = 2;
    T part[4] = {};
#pragma unroll
    for (int m0 = 0; m0 < runs; m0 += batch) {
        T entry[batch][8];
        T right[batch][8];
#pragma unroll
        for (int mb = 0; mb < batch; ++mb) {
            const int m = m0 + mb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * m + 8 * fk + j;
                // (columns beyond the row's last tile exist in LDS - the same
                // 64-column tile - and are masked out)
                T value{};
                if (m < runs) {
                    if (Tile < 4) {
                        value = x00[k * ld + row];
                    } else if (32 * m < H) {
                        value = x10[k * ld + row - H];
                    } else {
                        value = x11[(k - H) * ld + row - H];
                    }
                }
                entry[mb][j] =
                    (m < runs && (32 * m + 31 < columns || k < columns)) ? value
                                                                          : T{};
                right[mb][j] = m < runs ? rhs[k] : T{};
            }
        }
#pragma unroll
        for (int mb = 0; mb < batch; ++mb) {
            if (m0 + mb < runs) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    part[j & 3] += entry[mb][j] * right[mb][j];
                }
            }
        }
    }
    return (part[0] + part[1]) + (part[2] + part[3]);
}


}  // namespace detail
}  // namespace kernel


}  // namespace dev
}  // namespace accblas
