// TRSV kernels for gfx950: in-place x <- T^{-1} x, T the lower/upper triangle
// of a row-major matrix, unit or non-unit diagonal.
//
// Replaces `kernel::trsv_init`, `lower_trsv`, `upper_trsv`, `acc_lower_trsv`,
// `acc_upper_trsv` and the launchers `trsv` / `acc_trsv`
// (`/root/reference/cuda/trsv_kernels.cuh:38-42,69-235,263-432,455-488,
// 527-692,725-893,918-961`). Kept from the reference (after Hogg, "A Fast
// Dense Triangular Solve in CUDA"): ONE kernel, row blocks numbered by a
// ticket drawn at start (a block only ever waits on blocks that are already
// running, so no residency assumption), the diagonal tile inverted before the
// block has to wait, later blocks consuming earlier solutions as they appear.
// Re-designed for CDNA4 (the solve is a chain of n/64 dependent hops, so the
// design minimises what sits between "x_{b-1} published" and "x_b published"):
//   * tile = 64 x 64 (one wave = one tile row): half the hops of the
//     reference's 32-wide tiles;
//   * upper and lower are one kernel: upper is the lower problem of the
//     index-reversed system (i' = n-1-i), which reproduces the reference's
//     bottom-up block numbering with the ragged block at the top;
//   * hand-off = the data IS the flag (guide G16, recipe R2): every solved
//     element is published as one 8-byte {tag, value} granule with a single
//     write-through (sc1) store into a zeroed workspace; consumers poll the
//     granules they need with sc1 loads. No progress word, no drain, no fence
//     on the critical path (the reference's __threadfence + volatile spin,
//     `trsv_kernels.cuh:171-182,229-233`, would be 5 serialised memory round
//     trips per hop here); per-XCD L2s are not coherent and L1 is never
//     refreshed, so plain loads/stores are not an option;
//   * per block: all waves stage the diagonal tile AND the last off-diagonal
//     panel in LDS; wave 0 inverts the tile with each lane owning a column of
//     the inverse in registers (2016 FMAs, no synchronisation, instead of a
//     Gauss-Jordan sweep through shared memory with a sync per step);
//   * far panels (all but the last): 16-byte loads of the storage type,
//     double-buffered in registers together with the x granules of the next
//     panel, FMA in the arithmetic type, ONE cross-lane reduction at the end;
//   * the last panel is folded with the inverse ahead of time,
//     M = inv(T_bb) A_{b,b-1}, so that x_b = inv(T_bb) rhs_far - M x_{b-1}:
//     after x_{b-1} arrives ONE 64 x 64 mat-vec from LDS is left (lane = row,
//     columns split over the waves, vector entries broadcast with v_readlane,
//     per-wave partials combined through LDS, one barrier).
#pragma once

#include <hip/hip_runtime.h>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_tile = wave_size;  // 64
constexpr int trsv_waves = 8;         // waves per workgroup
constexpr int trsv_prefetch_depth = 2;  // far panels in flight per wave

using granule_word = unsigned long long;

/** 8-byte words of hand-off workspace needed per element of x. */
template <typename St>
constexpr int trsv_granule_words()
{
    return sizeof(St) == 8 ? 2 : 1;
}


// Diagnostic builds only (tools/tune_trsv.hip defines ACCBLAS_TRSV_DIAG): every
// block records 100 MHz wall-clock stamps of its phases into a buffer that no
// other code reads. The library is never built with it.
#ifdef ACCBLAS_TRSV_DIAG
__device__ unsigned long long *trsv_diag_buffer;
#define ACCBLAS_TRSV_STAMP(_block, _slot)                                  \
    do {                                                                   \
        if (lane_id() == 0 && trsv_diag_buffer != nullptr) {               \
            trsv_diag_buffer[(_block)*16 + (_slot)] = wall_clock64();       \
        }                                                                  \
    } while (false)
#else
#define ACCBLAS_TRSV_STAMP(_block, _slot) \
    do {                                  \
    } while (false)
#endif


namespace kernel {


/**
 * `kernel::trsv_init` (`cuda/trsv_kernels.cuh:38-42`) plus the re-arming of the
 * hand-off workspace: helper[0] = ~0 (stays ~0 unless a bounded spin expired,
 * then 0xdeadbeef), helper[1] = next ticket = 0, every granule word = 0
 * (tag 0 = not published).
 */
static __global__ __launch_bounds__(256) void trsv_init(
    std::uint32_t *helper, granule_word *granules, index_type num_words)
{
    const index_type tid = index_type{blockIdx.x} * blockDim.x + threadIdx.x;
    if (tid == 0) {
        helper[0] = ~std::uint32_t{0};
        helper[1] = 0;
    }
    for (index_type i = tid; i < num_words;
         i += index_type{gridDim.x} * blockDim.x) {
        granules[i] = 0;
    }
}


namespace detail {


constexpr granule_word granule_tag = granule_word{1} << 32;

// Every spin is bounded: a wave that polled `trsv_spin_limit` times without
// success (seconds; a healthy solve needs microseconds to milliseconds) writes
// `trsv_error_word` into helper[0] and gives up; every other poller sees that
// word within 1024 polls and gives up as well, so a protocol bug ends in a
// wrong result that the caller can detect (helper[0] != ~0u) instead of a hung
// GPU.
constexpr unsigned int trsv_spin_limit = 1u << 21;
constexpr std::uint32_t trsv_error_word = 0xdeadbeefu;

/** Returns true when the caller must stop waiting. */
__device__ __forceinline__ bool spin_expired(unsigned int &spins,
                                             std::uint32_t *helper)
{
    ++spins;
    if ((spins & 1023u) != 0) {
        return false;
    }
    const std::uint32_t state = __hip_atomic_load(
        helper, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (state == trsv_error_word) {
        return true;
    }
    if (spins >= trsv_spin_limit) {
        __hip_atomic_store(helper, trsv_error_word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

/** Value of `value` in lane `src_lane` (wave-uniform index), broadcast to all
 *  lanes through SGPRs (v_readlane_b32; two for 8-byte types). */
template <typename T>
__device__ __forceinline__ T read_lane(T value, int src_lane)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte values only");
    if (sizeof(T) == 4) {
        int bits;
        __builtin_memcpy(&bits, &value, 4);
        bits = __builtin_amdgcn_readlane(bits, src_lane);
        T out;
        __builtin_memcpy(&out, &bits, 4);
        return out;
    }
    int halves[2];
    __builtin_memcpy(halves, &value, 8);
    halves[0] = __builtin_amdgcn_readlane(halves[0], src_lane);
    halves[1] = __builtin_amdgcn_readlane(halves[1], src_lane);
    T out;
    __builtin_memcpy(&out, halves, 8);
    return out;
}

/** sum_{w < Count} values[w * stride + lane], combined as a balanced tree
 *  (fixed order, so the result is reproducible). */
template <int Count, typename T>
__device__ __forceinline__ T tree_sum(const T *values, int stride, int lane)
{
    static_assert(Count >= 1 && (Count & (Count - 1)) == 0, "power of two");
    T level[Count];
#pragma unroll
    for (int w = 0; w < Count; ++w) {
        level[w] = values[w * stride + lane];
    }
#pragma unroll
    for (int width = Count / 2; width >= 1; width /= 2) {
#pragma unroll
        for (int w = 0; w < width; ++w) {
            level[w] = level[2 * w] + level[2 * w + 1];
        }
    }
    return level[0];
}

/** Encoding of one storage element into 1 (<= 4 bytes) or 2 (8 bytes)
 *  granules: low 32 bits payload, high 32 bits tag. */
template <typename St>
struct granule_io {
    static constexpr int words = trsv_granule_words<St>();

    __device__ __forceinline__ static void publish(granule_word *granules,
                                                   index_type element, St value)
    {
        if (words == 1) {
            unsigned int bits = 0;
            __builtin_memcpy(&bits, &value, sizeof(St));
            __hip_atomic_store(granules + element, granule_tag | bits,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            granule_word bits = 0;
            __builtin_memcpy(&bits, &value, sizeof(St));
            __hip_atomic_store(granules + 2 * element,
                               granule_tag | (bits & 0xffffffffull),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(granules + 2 * element + 1,
                               granule_tag | (bits >> 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    __device__ __forceinline__ static void load(const granule_word *granules,
                                                index_type element,
                                                granule_word (&raw)[words])
    {
#pragma unroll
        for (int w = 0; w < words; ++w) {
            raw[w] = __hip_atomic_load(granules + words * element + w,
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    __device__ __forceinline__ static bool ready(
        const granule_word (&raw)[words])
    {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < words; ++w) {
            ok = ok && (raw[w] >> 32) == 1;
        }
        return ok;
    }

    __device__ __forceinline__ static St decode(
        const granule_word (&raw)[words])
    {
        St value;
        if (words == 1) {
            const unsigned int bits = static_cast<unsigned int>(raw[0]);
            __builtin_memcpy(&value, &bits, sizeof(St));
        } else {
            const granule_word bits =
                (raw[0] & 0xffffffffull) | (raw[words - 1] << 32);
            __builtin_memcpy(&value, &bits, sizeof(St));
        }
        return value;
    }
};


}  // namespace detail


/**
 * Blocked triangular solve. blockDim = 64 * Waves, grid = ceil(n / 64).
 *
 * Logical (always lower-triangular) indices i' map to memory indices by
 * `phys`: identity for Lower, n-1-i' for Upper. Logical block b covers
 * i' in [64 b, 64 b + 63]; entries with i' >= n do not exist (masked).
 * Granules are indexed by the logical element index.
 *
 * Vec = elements per 16-byte load in the far sweep (requires the matrix base
 * and row stride aligned to 16 bytes, and for Upper n % Vec == 0); Vec = 1 is
 * the fallback for any stride / alignment.
 *
 * Meaning per block as in `cuda/trsv_kernels.cuh:527-692` (lower) /
 * `:725-893` (upper): x_b = inv(T_bb) * (x_b - sum_{c<b} A_bc x_c), all in
 * ArType, x stored as StType and re-read by later blocks as stored.
 */
template <int Waves, bool Upper, bool UnitDiag, int Vec, typename ArType,
          typename MtxView, typename XView>
__global__ __launch_bounds__(Waves *wave_size) void trsv(
    MtxView mtx, XView x, std::uint32_t *helper, granule_word *granules)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using mtx_traits = view_traits<MtxView>;
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    using io = detail::granule_io<st_type>;
    constexpr int B = trsv_tile;
    constexpr int ld = B + 1;                 // LDS leading dimension
    constexpr int RW = B / Waves;             // rows per wave in the sweep
    constexpr int LPR = B / Vec;              // lanes sharing one tile row
    constexpr int NI = RW / Vec;              // load instructions per panel
    static_assert(RW % Vec == 0 && B % Vec == 0, "bad Vec");

    // tile: the diagonal tile row-major (tile[r*ld + c]), later its inverse
    // column-major (tile[c*ld + r]); panel: the last off-diagonal panel
    // column-major (panel[c*ld + r]) so that lane = row reads are conflict-free
    __shared__ ar_type tile[B * ld];
    __shared__ ar_type panel[B * ld];
    __shared__ ar_type correction[B];
    __shared__ ar_type partial_a[Waves * B];
    __shared__ ar_type partial_b[Waves * B];
    __shared__ ar_type x_prev_lds[B];
    __shared__ std::int32_t shared_block;

    const int lane = lane_id();
    const int wave = wave_id();
    const index_type n = mtx.length(0);

    if (threadIdx.x == 0) {
        shared_block = static_cast<std::int32_t>(
            __hip_atomic_fetch_add(helper + 1, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    const index_type block = shared_block;
    const index_type base = block * B;
    if (base >= n) {
        return;
    }
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 0);  // ticket drawn
    }
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };
    const st_type *a_data = mtx_traits::data(mtx);
    const index_type lda = mtx_traits::stride(mtx);
    st_type *x_data = x_traits::data(x);
    const index_type x_inc = x_traits::stride(x);

    // this block's right-hand side values (lane = row), never written by
    // another workgroup: plain load, issued early
    const index_type own_row = base + lane;
    const bool own_valid = own_row < n;
    st_type *own_addr = x_data + phys(own_valid ? own_row : n - 1) * x_inc;
    ar_type own_rhs{};
    if (own_valid) {
        own_rhs = static_cast<ar_type>(*own_addr);
    }

    // ---- diagonal tile and last panel -> LDS (a wave reads one row) -------
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
        if (block > 0) {
            ar_type off{};
            if (gr < n) {
                off = mtx(phys(gr), phys(base - B + lane));
            }
            panel[lane * ld + r] = off;
        }
    }
    __syncthreads();

    // ---- invert the tile: lane j computes column j of the inverse --------
    // X = T^{-1}: X[i][j] = (delta_ij - sum_{k<i} T[i][k] X[k][j]) / T[i][i]
    if (wave == 0) {
        ar_type column[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            // a dependent fp64 FMA chain costs ~40 cycles per link on gfx950:
            // four interleaved partial sums cut the row's latency by 4
            ar_type sums[4] = {};
#pragma unroll
            for (int k = 0; k < i; ++k) {
                sums[k & 3] += tile[i * ld + k] * column[k];  // LDS broadcast
            }
            ar_type value = ((lane == i) ? ar_type{1} : ar_type{0}) -
                            ((sums[0] + sums[1]) + (sums[2] + sums[3]));
            if (!UnitDiag) {
                value = value / tile[i * ld + i];
            }
            column[i] = value;
        }
        // all reads of T are done (same wave, program order): overwrite with
        // the inverse, column-major
#pragma unroll
        for (int i = 0; i < B; ++i) {
            tile[lane * ld + i] = column[i];
        }
        ACCBLAS_TRSV_STAMP(block, 1);  // inverse ready
    }
    __syncthreads();

    // ---- fold the inverse into the last panel: M = inv(T_bb) * A_{b,b-1} ---
    // x_b = inv(T) (rhs_far - A_{b,b-1} x_{b-1}) = inv(T) rhs_far - M x_{b-1}:
    // with M prepared here, off the critical path, only ONE mat-vec is left
    // between the arrival of x_{b-1} and the publication of x_b. Each wave
    // computes CW columns of M (lane = row) and overwrites the same columns of
    // `panel`; the lane's row of the inverse is held in registers.
    constexpr int CW = B / Waves;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (block > 0) {
        ar_type inv_row[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            inv_row[k] = tile[k * ld + lane];
        }
        constexpr int CG = 4;  // columns per pass: independent FMA chains
        static_assert(CW % CG == 0, "CW must be a multiple of 4");
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
            // every lane of this wave has read the CG columns (program order
            // within a wave; other waves never touch these columns)
#pragma unroll
            for (int q = 0; q < CG; ++q) {
                panel[(col + q) * ld + lane] = m[q];
            }
        }
    }
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 6);  // M ready
    }

    // ---- far sweep: correction = sum_{c < block-1} A[block][c] x_c --------
    // Instruction k of a wave covers tile rows wave*RW + k*Vec + lane/LPR and
    // Vec consecutive (logical) columns starting at (lane % LPR) * Vec.
    const index_type far_end = block - 1;  // panels [0, far_end)
    const int row_in_inst = lane / LPR;
    const int col_group = lane % LPR;
    struct stage {
        st_type a[NI][Vec];
        granule_word xraw[Vec][io::words];
    };
    const st_type *row_ptr[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const index_type gr = base + wave * RW + k * Vec + row_in_inst;
        row_ptr[k] = a_data + phys(gr < n ? gr : n - 1) * lda;
    }
    const auto issue = [&](index_type cb, stage &s) {
        const index_type first_col = cb * B + col_group * Vec;  // logical
        // lowest memory column of the group (columns descend for Upper)
        const index_type mem_col =
            Upper ? n - 1 - (first_col + Vec - 1) : first_col;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (Vec > 1) {
                const auto v = load_packed<(Vec > 1 ? Vec : 2), true>(
                    row_ptr[k] + mem_col);
#pragma unroll
                for (int e = 0; e < Vec; ++e) {
                    s.a[k][e] = v[e];
                }
            } else {
                s.a[k][0] = row_ptr[k][mem_col];
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
        const index_type first_col = cb * B + col_group * Vec;
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
                // memory element e of an Upper group is logical column
                // Vec-1-e of the group
                acc[k] += static_cast<ar_type>(s.a[k][e]) *
                          xv[Upper ? Vec - 1 - e : e];
            }
        }
    };
    if (far_end > 0) {
        // ring of `depth` stages: the matrix data of the next `depth` panels
        // is always in flight (it does not depend on x); each hop of the
        // front triggers one new 16 KB panel load per block, and its latency
        // under load (2-3 us) must not stall the sweep
        constexpr int depth = trsv_prefetch_depth;
        stage ring[depth];
#pragma unroll
        for (int d = 0; d < depth; ++d) {
            if (d < far_end) {
                issue(d, ring[d]);
            }
        }
        for (index_type cb0 = 0; cb0 < far_end; cb0 += depth) {
#pragma unroll
            for (int d = 0; d < depth; ++d) {
                const index_type cb = cb0 + d;
                if (cb < far_end) {
                    consume(cb, ring[d]);
                    if (cb + depth < far_end) {
                        issue(cb + depth, ring[d]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const ar_type total = group_all_reduce<LPR>(
            acc[k], [](ar_type a, ar_type b) { return a + b; });
        if (col_group == 0) {
            const int r = wave * RW + k * Vec + row_in_inst;
            correction[r] = (base + r < n) ? total : ar_type{};
        }
    }
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 2);  // wave 0 finished its far sweep share
    }
    __syncthreads();

    // ---- tail: c = inv(T) rhs_far (before the wait), then x = c - M x_{b-1} -
    // Both are 64 x 64 mat-vecs from LDS with lane = row. The columns are
    // split over the waves (CW each); a wave broadcasts its column's vector
    // entry from the lane that holds it with v_readlane (no LDS round trip),
    // and the per-wave partial results are combined through LDS in a fixed
    // order.
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 3);  // all waves finished the far sweep
    }
    ar_type rhs = own_valid ? own_rhs - correction[lane] : ar_type{};
    {
        // every wave holds the full right-hand side (lane = row); entry
        // wave*CW + j of it multiplies column wave*CW + j of the inverse
        ar_type part[4] = {};
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            part[j & 3] += tile[(wave_u * CW + j) * ld + lane] *
                           detail::read_lane(rhs, wave_u * CW + j);
        }
        partial_b[wave_u * B + lane] =
            (part[0] + part[1]) + (part[2] + part[3]);
    }
    if (block > 0) {
        // critical path starts here. ONE wave polls the 64 granules of
        // x_{b-1} (lane = element) and hands them to the others through LDS:
        // four independent pollers would each add their own poll-period skew
        // in front of the barrier, and their loads would queue beside the
        // critical one.
        if (wave == 0) {
            granule_word raw[io::words];
            const index_type prev = base - B + lane;  // always < n
            io::load(granules, prev, raw);
            while (!__all(io::ready(raw)) &&
                   !detail::spin_expired(spins, helper)) {
                io::load(granules, prev, raw);
            }
            ACCBLAS_TRSV_STAMP(block, 4);  // x_{b-1} observed
            x_prev_lds[lane] = static_cast<ar_type>(io::decode(raw));
        }
        __syncthreads();
        if (wave == 0) {
            ACCBLAS_TRSV_STAMP(block, 7);  // x_{b-1} handed to all waves
        }
        ar_type part[4] = {};
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            part[j & 3] += panel[(wave_u * CW + j) * ld + lane] *
                           x_prev_lds[wave_u * CW + j];
        }
        partial_a[wave_u * B + lane] =
            (part[0] + part[1]) + (part[2] + part[3]);
        if (wave == 0) {
            ACCBLAS_TRSV_STAMP(block, 8);  // wave 0 mat-vec share done
        }
    }
    __syncthreads();
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 9);  // all mat-vec shares done
        // balanced trees: a dependent fp64 add costs ~32 cycles on gfx950
        ar_type sum = detail::tree_sum<Waves>(partial_b, B, lane);
        if (block > 0) {
            sum -= detail::tree_sum<Waves>(partial_a, B, lane);
        }
        const st_type solution = static_cast<st_type>(sum);
        if (own_valid) {
            io::publish(granules, own_row, solution);  // hand-off first
            *own_addr = solution;                       // the user's vector
        }
        ACCBLAS_TRSV_STAMP(block, 5);  // published
    }
}


}  // namespace kernel


/** Bytes of hand-off workspace `launch_trsv` needs for an n-element solve. */
template <typename St>
constexpr std::size_t trsv_workspace_bytes(index_type n)
{
    return static_cast<std::size_t>(n) * trsv_granule_words<St>() *
           sizeof(granule_word);
}


template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv(bool upper, bool unit_diag, const MtxView &mtx,
                       const XView &x, std::uint32_t *helper,
                       granule_word *granules, hipStream_t stream)
{
    using mtx_traits = view_traits<MtxView>;
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    const index_type n = mtx.length(0);
    if (n <= 0) {
        return hipSuccess;
    }
    constexpr int waves = trsv_waves;
    constexpr int vec16 = 16 / sizeof(st_type);
    const dim3 grid(static_cast<unsigned>(ceildiv(n, trsv_tile)), 1, 1);
    const dim3 block(waves * wave_size, 1, 1);
    const index_type num_words = n * trsv_granule_words<st_type>();
    const unsigned init_blocks =
        static_cast<unsigned>(std::min<index_type>(ceildiv(num_words, 256), 512));
    kernel::trsv_init<<<init_blocks, 256, 0, stream>>>(helper, granules,
                                                       num_words);
    const bool vec_ok = is_aligned(mtx_traits::data(mtx), 16) &&
                        mtx_traits::stride(mtx) % vec16 == 0 &&
                        (!upper || n % vec16 == 0);
#define ACCBLAS_TRSV_LAUNCH(_upper, _unit)                                     \
    if (vec_ok) {                                                              \
        kernel::trsv<waves, _upper, _unit, vec16, ArType>                      \
            <<<grid, block, 0, stream>>>(mtx, x, helper, granules);            \
    } else {                                                                   \
        kernel::trsv<waves, _upper, _unit, 1, ArType>                          \
            <<<grid, block, 0, stream>>>(mtx, x, helper, granules);            \
    }
    if (upper) {
        if (unit_diag) {
            ACCBLAS_TRSV_LAUNCH(true, true)
        } else {
            ACCBLAS_TRSV_LAUNCH(true, false)
        }
    } else {
        if (unit_diag) {
            ACCBLAS_TRSV_LAUNCH(false, true)
        } else {
            ACCBLAS_TRSV_LAUNCH(false, false)
        }
    }
#undef ACCBLAS_TRSV_LAUNCH
    return hipGetLastError();
}


}  // namespace dev
}  // namespace accblas
