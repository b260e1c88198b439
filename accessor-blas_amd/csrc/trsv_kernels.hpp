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
// block has to wait, a "last finished block" word that later blocks poll.
// Re-designed for CDNA4:
//   * tile = 64 x 64 (one wave = one tile row): half the sequential hops of
//     the reference's 32-wide tiles;
//   * upper and lower are one kernel: upper is solved as the lower problem of
//     the index-reversed system (i' = n-1-i), which reproduces the reference's
//     bottom-up block numbering with the ragged block at the top;
//   * the tile inverse is computed by ONE wave with each lane owning a column
//     of the inverse in registers (2016 FMAs, no synchronisation), instead of
//     a Gauss-Jordan sweep through shared memory with a sync per step;
//   * waves wait on the progress word independently (no workgroup barrier in
//     the sweep) and only when their cached copy is too old;
//   * per-XCD L2s are not coherent and L1 is never refreshed: solved x values
//     are published with write-through (sc1) stores, drained, then the
//     progress word is stored; consumers poll it with sc1 loads and read x
//     with sc1 loads (guide G16, recipe R1). 2-byte storage uses an
//     agent-scope release/acquire fence pair instead (sc1 hand-offs are only
//     characterised for 4/8/16-byte accesses).
#pragma once

#include <hip/hip_runtime.h>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_tile = wave_size;  // 64
constexpr int trsv_waves = 4;         // waves per workgroup


namespace kernel {


/** `kernel::trsv_init` (`cuda/trsv_kernels.cuh:38-42`): helper[0] = index of
 *  the last finished block (all ones = none), helper[1] = next ticket. */
__global__ __launch_bounds__(64) void trsv_init(std::uint32_t *helper)
{
    if (threadIdx.x == 0) {
        helper[0] = ~std::uint32_t{0};
        helper[1] = 0;
    }
}


namespace detail {


template <typename St>
struct bits_of {
    using type = std::conditional_t<
        sizeof(St) == 2, unsigned short,
        std::conditional_t<sizeof(St) == 4, unsigned int, unsigned long long>>;
};

/** Write-through store of one storage element (visible to other XCDs after
 *  the storing wave drained its vector memory queue). */
template <typename St>
__device__ __forceinline__ void publish_store(St *addr, St value)
{
    using bits_type = typename bits_of<St>::type;
    bits_type bits;
    __builtin_memcpy(&bits, &value, sizeof(St));
    __hip_atomic_store(reinterpret_cast<bits_type *>(addr), bits,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/** L1-bypassing load of one storage element another workgroup published. */
template <typename St>
__device__ __forceinline__ St consume_load(const St *addr)
{
    using bits_type = typename bits_of<St>::type;
    const bits_type bits = __hip_atomic_load(
        reinterpret_cast<const bits_type *>(addr), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    St value;
    __builtin_memcpy(&value, &bits, sizeof(St));
    return value;
}


}  // namespace detail


/**
 * Blocked triangular solve. blockDim = (64 * Waves), grid = ceil(n / 64).
 *
 * Logical (always lower-triangular) indices i' are mapped to memory indices
 * by `phys`: identity for Lower, n-1-i' for Upper. Logical block b covers
 * i' in [64 b, 64 b + 63]; entries with i' >= n do not exist (masked).
 *
 * Meaning per block as in `cuda/trsv_kernels.cuh:527-692` (lower) /
 * `:725-893` (upper): x_b = inv(T_bb) * (x_b - sum_{c<b} A_bc x_c), all in
 * ArType, x stored as StType.
 */
template <int Waves, bool Upper, bool UnitDiag, typename ArType,
          typename MtxView, typename XView>
__global__ __launch_bounds__(Waves *wave_size) void trsv(
    MtxView mtx, XView x, std::uint32_t *helper)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using x_traits = view_traits<XView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    constexpr int B = trsv_tile;
    constexpr int ld = B + 1;                  // LDS leading dimension
    constexpr int rows_per_wave = B / Waves;   // rows each wave sweeps
    constexpr bool fenced = sizeof(st_type) == 2;

    // tile: first the diagonal tile row-major (tile[r*ld + c]), later its
    // inverse column-major (tile[c*ld + r])
    __shared__ ar_type tile[B * ld];
    __shared__ ar_type correction[B];
    __shared__ ar_type rhs[B];
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
    const auto phys = [n](index_type logical) {
        return Upper ? n - 1 - logical : logical;
    };
    st_type *x_data = x_traits::data(x);
    const index_type x_inc = x_traits::stride(x);

    // ---- diagonal tile -> LDS (coalesced: a wave reads one row) ----------
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
    __syncthreads();

    // ---- invert the tile: lane j computes column j of the inverse --------
    // X = T^{-1}: X[i][j] = (delta_ij - sum_{k<i} T[i][k] X[k][j]) / T[i][i]
    if (wave == 0) {
        ar_type column[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            ar_type value = (lane == i) ? ar_type{1} : ar_type{0};
#pragma unroll
            for (int k = 0; k < i; ++k) {
                value -= tile[i * ld + k] * column[k];  // LDS broadcast read
            }
            if (!UnitDiag) {
                value = value / tile[i * ld + i];
            }
            column[i] = value;
        }
        // all reads of T are done (same wave, program order): overwrite with
        // the inverse, column-major so that lane = row reads are conflict-free
#pragma unroll
        for (int i = 0; i < B; ++i) {
            tile[lane * ld + i] = column[i];
        }
    }

    // ---- off-diagonal sweep: correction = sum_{c<block} A[block][c] x_c ---
    ar_type partial[rows_per_wave];
#pragma unroll
    for (int l = 0; l < rows_per_wave; ++l) {
        partial[l] = ar_type{};
    }
    // rows of this wave: wave, wave + Waves, ... ; clamp masked rows for
    // loading and zero their contribution
    std::int32_t known_done = -1;
    for (index_type cb = 0; cb < block; ++cb) {
        if (known_done < cb) {
            // one relaxed sc1 poll per wave; uniform result
            std::int32_t seen;
            do {
                seen = static_cast<std::int32_t>(__hip_atomic_load(
                    helper, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                seen = __builtin_amdgcn_readfirstlane(seen);
                if (seen < cb) {
                    __builtin_amdgcn_s_sleep(1);
                }
            } while (seen < cb);
            known_done = seen;
            if (fenced) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                // keeps the compiler from hoisting the x load over the poll
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        const index_type gc = cb * B + lane;
        const ar_type x_value = static_cast<ar_type>(
            detail::consume_load(x_data + phys(gc) * x_inc));
#pragma unroll
        for (int l = 0; l < rows_per_wave; ++l) {
            const index_type gr = base + wave + l * Waves;
            if (gr < n) {
                partial[l] += x_value * mtx(phys(gr), phys(gc));
            }
        }
    }
#pragma unroll
    for (int l = 0; l < rows_per_wave; ++l) {
        const ar_type total = group_all_reduce(
            partial[l], [](ar_type a, ar_type b) { return a + b; });
        if (lane == 0) {
            correction[wave + l * Waves] = total;
        }
    }
    __syncthreads();

    // ---- solve the diagonal system with the inverse (lane = row) ----------
    if (wave == 0) {
        const index_type gr = base + lane;
        const bool valid = gr < n;
        st_type *x_addr = x_data + phys(valid ? gr : n - 1) * x_inc;
        ar_type local_rhs{};
        if (valid) {
            local_rhs = static_cast<ar_type>(*x_addr) - correction[lane];
        }
        rhs[lane] = local_rhs;
        ar_type sum[4] = {};
#pragma unroll
        for (int c = 0; c < B; c += 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sum[s] += tile[(c + s) * ld + lane] * rhs[c + s];
            }
        }
        const ar_type solution = (sum[0] + sum[1]) + (sum[2] + sum[3]);
        if (valid) {
            detail::publish_store(x_addr, static_cast<st_type>(solution));
        }
        drain_vmem();
        if (fenced) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_vmem();
        }
        if (lane == 0) {
            __hip_atomic_store(helper, static_cast<std::uint32_t>(block),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


}  // namespace kernel


template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv(bool upper, bool unit_diag, const MtxView &mtx,
                       const XView &x, std::uint32_t *helper,
                       hipStream_t stream)
{
    const index_type n = mtx.length(0);
    if (n <= 0) {
        return hipSuccess;
    }
    constexpr int waves = trsv_waves;
    const dim3 grid(static_cast<unsigned>(ceildiv(n, trsv_tile)), 1, 1);
    const dim3 block(waves * wave_size, 1, 1);
    kernel::trsv_init<<<1, 64, 0, stream>>>(helper);
    if (upper) {
        if (unit_diag) {
            kernel::trsv<waves, true, true, ArType>
                <<<grid, block, 0, stream>>>(mtx, x, helper);
        } else {
            kernel::trsv<waves, true, false, ArType>
                <<<grid, block, 0, stream>>>(mtx, x, helper);
        }
    } else {
        if (unit_diag) {
            kernel::trsv<waves, false, true, ArType>
                <<<grid, block, 0, stream>>>(mtx, x, helper);
        } else {
            kernel::trsv<waves, false, false, ArType>
                <<<grid, block, 0, stream>>>(mtx, x, helper);
        }
    }
    return hipGetLastError();
}


}  // namespace dev
}  // namespace accblas
