// Host side of the 128-row TRSV kernel (`kernel::trsv_wide`,
// trsv_wide_kernels.hpp): how a solve is split into tickets, which solves get
// prefetching workgroups, and the launchers. Included at the end of
// trsv_wide_kernels.hpp.
#pragma once

#include "trsv_wide_kernels.hpp"


namespace accblas {
namespace dev {


/** Dynamic LDS on top of the static arrays so that ONE workgroup is resident
 *  per CU (see `trsv_single_block_pad`); fp64 arithmetic needs none (136 KiB
 *  of static LDS). */
template <typename ArType>
constexpr unsigned trsv_wide_single_block_pad()
{
    constexpr std::size_t H = trsv_tile;
    constexpr std::size_t static_bytes =
        sizeof(ArType) * (4 * H * (H + 1) + 3 * trsv_wide_rows) + 16;
    constexpr std::size_t needed = 160 * 1024 / 2 + 2048;
    return static_bytes >= needed
               ? 0u
               : static_cast<unsigned>(needed - static_bytes);
}


/** Operands the 128-row kernel takes: those whose rows the sweeps may read
 *  with 16-byte loads (`trsv_vector_rows`: any since round 4; only aligned
 *  bases and row strides with `any_alignment` = false). */
template <typename MtxView>
bool trsv_wide_applicable(bool upper, const MtxView &mtx,
                          bool any_alignment = true)
{
    (void)upper;  // (upper solves of any n: mirrored around a virtual size)
    return trsv_vector_rows(mtx, any_alignment);
}



/** The split of a large solve (`trsv_wide_split_for`; how each value was
 *  arrived at: the comment there). */
constexpr int trsv_wide_parts = 5;             // workgroups per split row block
constexpr int trsv_wide_tail_parts = 8;        // ... of the tail blocks
constexpr int trsv_wide_tail_blocks = 32;      // tail: at least this many blocks
constexpr int trsv_wide_max_tail_blocks = 96;  // ... at most this many
constexpr int trsv_wide_last_parts = 16;       // last third of the tail
constexpr int trsv_wide_tail_deferred = 16;    // columns a tail owner keeps
constexpr int trsv_wide_whole_blocks = 32;     // blocks beyond the CUs kept whole, at most

/** How a solve of n unknowns is split on a device with `num_cus` CUs: row
 *  blocks that cannot be resident from the start (more blocks than CUs) get
 *  trsv_wide_parts workgroups each (4 helpers + the owner), the last
 *  trsv_wide_tail_blocks of them trsv_wide_tail_parts, their
 *  owners keeping trsv_wide_tail_deferred block columns only: the
 *  workgroups with the last tickets run almost alone, so their shares decide
 *  how long the solve's tail is. Measured (profiles/r02_trsv_split_ab.txt,
 *  us, unsplit -> 5 parts -> 5 parts + 8 for the last 64 blocks):
 *  n = 40000: 845-863 -> 747-785 -> 748-778; n = 65536: 1554 -> 1563-1608 ->
 *  1496-1500; the other tail shapes tried since are within 2 % of each other
 *  (profiles/r02_trsv_split_first.txt).
 *  Round 3: the solve's drain - the time between the last ticket and the last
 *  publication, CUs going idle one by one - is as long as the last shares take
 *  at what ONE workgroup streams (39 GB/s: 4.5 MB shares, 110-130 us at
 *  n = 65536, profiles/r03_trsv_tail.txt). Large solves therefore get a longer
 *  tail (all split blocks but the first 32, up to
 *  trsv_wide_max_tail_blocks) whose last third has
 *  trsv_wide_last_parts workgroups per block: 65536 -1.5...-3 %, 57344
 *  -1...-1.7 %, 40000-49152 unchanged within +-0.3 %; 24 or 32 parts for the
 *  last blocks are no better than 16 (every part costs its owner a wait). */
inline trsv_wide_split trsv_wide_split_for(index_type n, int num_cus,
                                           int split_from = 0)
{
    const int blocks = static_cast<int>(ceildiv(n, index_type{trsv_wide_rows}));
    // a split block must defer its fold (2 x deferred + 1 far columns at least)
    constexpr int min_first = 2 * static_cast<int>(trsv_wide_deferred_columns) + 2;
    // The first blocks beyond the CU count stay whole: the owner of a split
    // block starts when the front is still far away (a CU frees up per hop AND
    // per finished helper) and then holds its CU idle, a whole block that
    // starts early sweeps its own columns meanwhile. How many: round 2 kept half
    // of what there is, up to 64 (profiles/r02_trsv_split_first.txt: first
    // split block 256 -> 320 at n = 49152 / 57344 / 65536: 965 -> 922, 1190 ->
    // 1128, 1447 -> 1414 us). Re-measured in round 5 with the front loop and
    // without deferred folds for resident blocks (tools/tune_trsv_split.hip,
    // all candidates alternated in one process, profiles/r05_trsv_split_first.txt):
    // a sixth of what there is, up to 32 - first split block 288 instead of 320
    // at n = 57344: 1143 -> 1078 us, 272 at 49152: 907 -> 866, 272 at 45056:
    // 774 -> 735, 264 at 40000: 670 -> 656; 65536 is flat from 272 to 320
    // (1360-1366), 36000 from 258 to 274.
    int whole = blocks > num_cus ? (blocks - num_cus) / 6 : 0;
    whole = whole < trsv_wide_whole_blocks ? whole : trsv_wide_whole_blocks;
    // (`split_from` > 0: the caller's choice of the first split block)
    const int wanted = split_from > 0 ? split_from : num_cus + whole;
    const int first = wanted > min_first ? wanted : min_first;
    if (trsv_wide_parts > 1 && blocks > first) {
        const int split_blocks = blocks - first;
        int tail_blocks = split_blocks - trsv_wide_tail_blocks;
        tail_blocks = tail_blocks > trsv_wide_max_tail_blocks
                          ? trsv_wide_max_tail_blocks
                          : tail_blocks;
        tail_blocks = tail_blocks < trsv_wide_tail_blocks
                          ? trsv_wide_tail_blocks
                          : tail_blocks;
        tail_blocks = tail_blocks < split_blocks ? tail_blocks : split_blocks;
        const int last_blocks =
            trsv_wide_last_parts > trsv_wide_tail_parts
                ? tail_blocks / 3
                : 0;
        return trsv_wide_split{first,
                               trsv_wide_parts,
                               blocks - tail_blocks,
                               trsv_wide_tail_parts,
                               trsv_wide_tail_deferred,
                               blocks - last_blocks,
                               last_blocks > 0 ? trsv_wide_last_parts : 1,
                               0,
                               0,
                               0};
    }
    return trsv_wide_split{blocks, 1, blocks, 1,
                           static_cast<int>(trsv_wide_deferred_columns), blocks,
                           1, 0, 0, 0};
}

/** Tickets (units of work: row blocks, helpers' shares, prefetchers) of a
 *  solve under `split`. */
inline unsigned trsv_wide_tickets(index_type n, const trsv_wide_split &split)
{
    const index_type blocks = ceildiv(n, index_type{trsv_wide_rows});
    return static_cast<unsigned>(
        split.first_block + (split.tail_block - split.first_block) * split.parts +
        (split.last_block - split.tail_block) * split.tail_parts +
        (blocks - split.last_block) * split.last_parts + split.prefetchers);
}

/** The 128-row kernel under an explicit `split` (tools; the library goes
 *  through `launch_trsv_wide`). */
template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv_wide_split(bool upper, bool unit_diag,
                                  const MtxView &mtx, const XView &x,
                                  std::uint32_t *helper,
                                  const trsv_session &session,
                                  hipStream_t stream,
                                  const trsv_wide_split &split_in,
                                  int num_cus = 0, int persistent_mode = 0)
{
    trsv_wide_split split = split_in;
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    static_assert(view_traits<MtxView>::is_pointer_view &&
                      view_traits<XView>::is_pointer_view,
                  "reduced_row_major / row_major / plain views only");
    static_assert(trsv_wide_parts <= trsv_wide_max_parts &&
                      trsv_wide_tail_parts <= trsv_wide_max_parts &&
                      trsv_wide_last_parts <= trsv_wide_max_parts,
                  "too many parts");
    if (mtx.length(0) <= 0) {
        return hipSuccess;
    }
    // (the size the row blocks are laid out for, `trsv_wide_virtual_pad`)
    const index_type n =
        mtx.length(0) + trsv_wide_virtual_pad<st_type>(upper, mtx.length(0));
    if (split.parts > trsv_wide_max_parts ||
        split.tail_parts > trsv_wide_max_parts || split.parts < 1 ||
        split.tail_parts < 1 || split.last_parts > trsv_wide_max_parts ||
        split.last_parts < 1 || split.prefetchers < 0 ||
        split.last_block > ceildiv(n, index_type{trsv_wide_rows}) ||
        split.last_block < split.tail_block) {
        return hipErrorInvalidValue;
    }
    if (split.parts == 1 && (split.tail_parts != 1 || split.last_parts != 1)) {
        // (no splitting means none at all: the grid would hold workgroups
        // that no row block is waiting for)
        return hipErrorInvalidValue;
    }
    if (split.parts > 1) {
        // a split block adds its helpers' sums where it folds its panel, which
        // it only does late (`defer_fold`) when it has at least
        // 2 x deferred + 1 far columns; and its owner keeps `deferred` columns,
        // which it must have
        constexpr int min_first =
            2 * static_cast<int>(trsv_wide_deferred_columns) + 2;
        if (split.first_block < min_first ||
            split.tail_block < split.first_block || split.tail_deferred < 0 ||
            split.tail_deferred > split.tail_block - 1) {
            return hipErrorInvalidValue;
        }
    }
    constexpr int vec16 = 16 / sizeof(st_type);
    // a sweep step of 8-byte storage covers 64 columns: 16 load instructions
    // per lane and step would not leave registers for two steps in flight
    constexpr int panel_cols = sizeof(st_type) == 8 ? 64 : 128;
    // Steps of matrix data in flight per wave. While the folded panel W (64
    // registers with fp64 arithmetic) is live, a workgroup is at the chain's
    // front or within 48 columns of its own diagonal - it gets one column per
    // hop, two steps are plenty - and that is also what the 256 registers of
    // a lane hold next to W. The bulk of a long sweep runs before W exists
    // (`defer_fold`, helpers of split blocks): as deep as the registers allow.
    constexpr int depth = sizeof(st_type) == 2 ? 3 : 2;
    constexpr int deep_depth = sizeof(st_type) == 2 ? 5 : 4;
    const unsigned tickets = trsv_wide_tickets(n, split);
    split.tickets = static_cast<int>(tickets);
    // (row blocks below the CU count start with the solve and fold at once)
    split.defer_from = num_cus > 0 ? num_cus : 0;
    // (one workgroup per CU drawing tickets when there are more tickets than
    // CUs; a launch with a workgroup for every ticket runs the loop-free kernel)
    const bool persistent = persistent_mode != 0 && num_cus > 0 &&
                            tickets > static_cast<unsigned>(num_cus);
    const dim3 grid(persistent ? static_cast<unsigned>(num_cus) : tickets);
    const dim3 block(trsv_wide_waves * wave_size);
    const unsigned pad = trsv_wide_single_block_pad<ArType>();
#define ACCBLAS_TRSV_WIDE_LAUNCH(_upper, _unit)                                \
    if (persistent) {                                                          \
        kernel::trsv_wide<_upper, _unit, vec16, panel_cols, depth, deep_depth, \
                          ArType, true>                                        \
            <<<grid, block, pad, stream>>>(mtx, x, helper, session, split);    \
    } else {                                                                   \
        kernel::trsv_wide<_upper, _unit, vec16, panel_cols, depth, deep_depth, \
                          ArType, false>                                       \
            <<<grid, block, pad, stream>>>(mtx, x, helper, session, split);    \
    }
    if (upper) {
        if (unit_diag) {
            ACCBLAS_TRSV_WIDE_LAUNCH(true, true);
        } else {
            ACCBLAS_TRSV_WIDE_LAUNCH(true, false);
        }
    } else {
        if (unit_diag) {
            ACCBLAS_TRSV_WIDE_LAUNCH(false, true);
        } else {
            ACCBLAS_TRSV_WIDE_LAUNCH(false, false);
        }
    }
#undef ACCBLAS_TRSV_WIDE_LAUNCH
    return hipGetLastError();
}

/** Prefetching workgroups a solve gets (`trsv_wide_split::prefetchers`): the
 *  CUs the row blocks leave idle, up to 128 of them, where it was measured to
 *  pay. Round 3 (profiles/r03_trsv_prefetch.txt): 4-byte storage -3...-6 % and
 *  8-byte storage -12...-26 % at 32 to 144 row blocks, 2-byte storage +5 %.
 *  Round 5, re-measured under round 4's front loop (the followers' tile
 *  requests now stay in flight across a hop, so a tile that waits in the
 *  Infinity Cache buys nothing and its second trip through the fabric costs):
 *  4-byte storage is 2-5 % SLOWER with prefetchers at n = 4096 ... 12288 and
 *  8-25 % slower at 14336 ... 28000, both triangles, fp64 and fp32 arithmetic
 *  (profiles/r05_trsv_prefetch_ab.txt) - the library's sweep had a 15 % cliff at
 *  144 row blocks where the old rule switched them off. The rule is now 8-byte
 *  storage only (which the library sends to the 64-row kernel unless option
 *  "trsv_wide" = 2 asks for this one). `mode`: 0 never, 1 by that rule,
 *  2 whenever CUs are idle (measurements). */
template <typename StType>
inline int trsv_wide_prefetchers_for(index_type n, int num_cus, int mode)
{
    const int blocks = static_cast<int>(ceildiv(n, index_type{trsv_wide_rows}));
    const int spare = num_cus - blocks;
    if (mode == 0 || spare <= 0) {
        return 0;
    }
    if (mode == 1 && (sizeof(StType) < 8 || blocks < 32 || blocks > 144)) {
        return 0;
    }
    return spare < 128 ? spare : 128;
}

template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv_wide(bool upper, bool unit_diag, const MtxView &mtx,
                            const XView &x, std::uint32_t *helper,
                            const trsv_session &session, hipStream_t stream,
                            int num_cus, int split_from = 0,
                            int prefetch_mode = 1, int persistent_mode = 0)
{
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    const index_type laid_out =
        mtx.length(0) + trsv_wide_virtual_pad<st_type>(upper, mtx.length(0));
    trsv_wide_split split = trsv_wide_split_for(laid_out, num_cus, split_from);
    split.prefetchers =
        split.parts == 1 ? trsv_wide_prefetchers_for<st_type>(
                               laid_out, num_cus, prefetch_mode)
                         : 0;
    return launch_trsv_wide_split<ArType>(upper, unit_diag, mtx, x, helper,
                                          session, stream, split, num_cus,
                                          persistent_mode);
}


}  // namespace dev
}  // namespace accblas
