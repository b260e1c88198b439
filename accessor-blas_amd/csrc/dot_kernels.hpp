// DOT kernels for gfx950: *res = sum_i x(i,0) * y(i,0).
//
// Replaces `kernel::init_res`, `kernel::dot`, `kernel::acc_dot`,
// `kernel::cast_result` and the launchers `dot` / `acc_dot`
// (`/root/reference/cuda/dot_kernels.cuh:78-173,192-263`). The reference needs
// up to three launches (zero the result, grid-stride multiply-add with one
// atomic per block, cast) and its result depends on the atomic arrival order.
// Here the default is ONE launch with a reproducible result:
//   * persistent grid (CUs x blocks_per_cu), 16-byte loads of both vectors in
//     the storage type, `Unroll` pairs of loads in flight per lane, widen to
//     the arithmetic type in registers;
//   * lanes -> wave by shuffle, waves -> workgroup through LDS, then each
//     workgroup writes its partial to a workspace slot (sc1 store), drains,
//     and takes a ticket; the workgroup that draws the last ticket adds the
//     partials in slot order and writes the result already cast to ResType
//     (so `init_res` and `cast_result` disappear);
//   * the hand-off follows the guide's agent-scope recipe (sc1 payload,
//     s_waitcnt vmcnt(0), agent-scope atomic ticket, sc1 reads after the
//     ticket returned) and is independent of dispatch order and XCD placement.
// ACCBLAS_DOT_ATOMIC keeps the reference's behaviour (atomic_add per
// workgroup) for comparison.
#pragma once

#include <hip/hip_runtime.h>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


constexpr int dot_max_partials = 4096;    // workspace slots in the handle
constexpr int dot_ticket_shards = 16;     // arrival counters, one 64 B line each
constexpr int dot_ticket_words = (dot_ticket_shards + 1) * 16;  // + top counter


/** Tuning knobs: waves per workgroup, independent load pairs per lane,
 *  workgroups per CU of the persistent grid, and whether a workgroup sweeps a
 *  contiguous slice of the vectors (true) or the whole grid strides together
 *  (false). */
template <int Waves, int Unroll, int BlocksPerCu, bool Contiguous>
struct dot_config {
    static constexpr int waves = Waves;
    static constexpr int unroll = Unroll;
    static constexpr int blocks_per_cu = BlocksPerCu;
    static constexpr bool contiguous = Contiguous;
};

/** Production shapes, chosen by measurement on MI355X (tools/tune_dot.hip,
 *  profiles/r01_tune_dot.txt, profiles/r02_tune_dot.txt): 512-thread
 *  workgroups. Long streams run fastest when each workgroup sweeps one
 *  contiguous slice with ONE workgroup per CU and 8 load pairs per lane in
 *  flight (256 KB per CU; 2 x 1 GiB fp32: 306 vs 310 us with two workgroups of
 *  4 pairs, 1 GiB of fp16: 156 vs 159 us and 3 % better in the median);
 *  shorter ones when the whole grid strides together with 2 workgroups per CU
 *  and 4 pairs. The two cross between 256 and 512 MiB of data
 *  (profiles/r02_tune_dot_rule.txt, us, strided / contiguous: 128 MiB of fp16 -
 *  the per-GPU share of config C4 on 8 GPUs - 25.2 / 30.5, 256 MiB 45.6 /
 *  46.8, 384 MiB 66.4 / 66.2, 512 MiB 85.3 / 81.1, 1 GiB 166 / 164; fp32
 *  storage 23.9 / 24.9, 43.1 / 44.2, 66.4 / 65.2, 87.3 / 80.7, 172 / 155). */
using dot_stride_config = dot_config<8, 4, 2, false>;
using dot_contiguous_config = dot_config<8, 8, 1, true>;
constexpr index_type dot_contiguous_min_bytes = index_type{3} << 27;  // 384 MiB
struct dot_auto_config {};


namespace kernel {


/** `kernel::init_res` (`cuda/dot_kernels.cuh:78-82`), atomic mode only. */
template <typename ValueType>
__global__ __launch_bounds__(64) void init_res(ValueType *res)
{
    if (threadIdx.x == 0) {
        *res = ValueType{0};
    }
}

/** `kernel::cast_result` (`cuda/dot_kernels.cuh:168-173`), atomic mode only. */
template <typename InType, typename OutType>
__global__ __launch_bounds__(64) void cast_result(const InType *in,
                                                  OutType *out)
{
    if (threadIdx.x == 0) {
        *out = static_cast<OutType>(*in);
    }
}


/**
 * Grid-stride dot product. Vec > 1 requires both storage pointers aligned to
 * Vec*sizeof(St) and unit increments (checked by the launcher); Vec == 1 goes
 * through operator() only and works for any increment.
 *
 * Atomic == false: partials[blockIdx.x] + ticket protocol, result written by
 * the last workgroup as ResType. Atomic == true: atomic_add(accum, partial)
 * as `cuda/dot_kernels.cuh:158-160` (accum must be zeroed before).
 */
template <typename Config, int Vec, bool Atomic, typename XView,
          typename YView, typename ArType, typename ResType>
__global__ __launch_bounds__(Config::waves *wave_size) void dot(
    XView x, YView y, ArType *partials, unsigned int *ticket, ResType *res)
{
    constexpr int Waves = Config::waves;
    constexpr int Unroll = Config::unroll;
    using ar_type = decltype(x(0, 0) + y(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
    using x_traits = view_traits<XView>;
    using y_traits = view_traits<YView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    constexpr int block_threads = Waves * wave_size;

    const index_type n = x.length(0);
    const index_type tid =
        index_type{blockIdx.x} * block_threads + threadIdx.x;
    // grid-stride: all threads of the grid advance together; contiguous: each
    // workgroup owns one slice [first, last) of the chunk range
    const index_type num_threads =
        Config::contiguous ? index_type{block_threads}
                           : index_type{gridDim.x} * block_threads;

    ar_type acc[Unroll];
#pragma unroll
    for (int u = 0; u < Unroll; ++u) {
        acc[u] = ar_type{};
    }

    if constexpr (Vec > 1) {
        const st_type *x_ptr = x_traits::data(x);
        const st_type *y_ptr = y_traits::data(y);
        const index_type all_chunks = n / Vec;
        // slices start on 4 KiB boundaries (256 chunks of 16 bytes)
        const index_type per_block =
            ceildiv(ceildiv(all_chunks, gridDim.x), 256) * 256;
        const index_type first =
            Config::contiguous ? index_type{blockIdx.x} * per_block : 0;
        const index_type num_chunks =
            Config::contiguous
                ? (first + per_block < all_chunks
                       ? first + per_block
                       : (first < all_chunks ? all_chunks : first))
                : all_chunks;
        index_type chunk = Config::contiguous ? first + threadIdx.x : tid;
        // Unroll independent chunk pairs per lane per iteration
        for (; chunk + (Unroll - 1) * num_threads < num_chunks;
             chunk += Unroll * num_threads) {
            typename packed<st_type, (Vec > 1 ? Vec : 2)>::type xv[Unroll];
            typename packed<st_type, (Vec > 1 ? Vec : 2)>::type yv[Unroll];
#pragma unroll
            for (int u = 0; u < Unroll; ++u) {
                const index_type c = (chunk + u * num_threads) * Vec;
                xv[u] = load_packed<(Vec > 1 ? Vec : 2), true>(x_ptr + c);
                yv[u] = load_packed<(Vec > 1 ? Vec : 2), true>(y_ptr + c);
            }
#pragma unroll
            for (int u = 0; u < Unroll; ++u) {
#pragma unroll
                for (int k = 0; k < Vec; ++k) {
                    acc[u] += static_cast<ar_type>(xv[u][k]) *
                              static_cast<ar_type>(yv[u][k]);
                }
            }
        }
        for (; chunk < num_chunks; chunk += num_threads) {
            const index_type c = chunk * Vec;
            const auto xv = load_packed<(Vec > 1 ? Vec : 2), true>(x_ptr + c);
            const auto yv = load_packed<(Vec > 1 ? Vec : 2), true>(y_ptr + c);
#pragma unroll
            for (int k = 0; k < Vec; ++k) {
                acc[0] += static_cast<ar_type>(xv[k]) *
                          static_cast<ar_type>(yv[k]);
            }
        }
        // elements past the last whole chunk (< Vec of them)
        const index_type tail = all_chunks * Vec + tid;
        if (tail < n) {
            acc[0] += x(tail, 0) * y(tail, 0);
        }
    } else {
        const index_type grid_threads = index_type{gridDim.x} * block_threads;
        index_type idx = tid;
        for (; idx + (Unroll - 1) * grid_threads < n;
             idx += Unroll * grid_threads) {
#pragma unroll
            for (int u = 0; u < Unroll; ++u) {
                acc[u] += x(idx + u * grid_threads, 0) *
                          y(idx + u * grid_threads, 0);
            }
        }
        for (; idx < n; idx += grid_threads) {
            acc[0] += x(idx, 0) * y(idx, 0);
        }
    }

    ar_type local = acc[0];
#pragma unroll
    for (int u = 1; u < Unroll; ++u) {
        local += acc[u];
    }

    __shared__ ar_type scratch[Waves];
    __shared__ bool is_last;
    const ar_type block_sum = block_reduce_sum<Waves>(local, scratch);

    if (Atomic) {
        if (threadIdx.x == 0) {
            atomic_add(partials, block_sum);
        }
        return;
    }

    // publish the partial, then take a ticket (guide G16: sc1 payload ->
    // drain -> agent-scope atomic; the last arriver is told by the value its
    // add returned). Same-address atomics retire at ~12 ns each, so arrivals
    // are spread over dot_ticket_shards counters on separate 64 B lines; the
    // last arriver of a shard then arrives at the top counter.
    if (threadIdx.x == 0) {
        store_agent(partials + blockIdx.x, block_sum);
        drain_vmem();
        const unsigned int shard = blockIdx.x % dot_ticket_shards;
        const unsigned int shard_size =
            (gridDim.x - shard + dot_ticket_shards - 1) / dot_ticket_shards;
        const unsigned int active_shards =
            gridDim.x < dot_ticket_shards ? gridDim.x : dot_ticket_shards;
        bool last = false;
        const unsigned int drawn = __hip_atomic_fetch_add(
            ticket + 16 * (shard + 1), 1u, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
        if (drawn == shard_size - 1) {
            const unsigned int top = __hip_atomic_fetch_add(
                ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (top == active_shards - 1);
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) {
        return;
    }
    // last workgroup: fixed-order sum of the partials (sc1 loads)
    ar_type total{};
    for (unsigned int slot = threadIdx.x; slot < gridDim.x;
         slot += block_threads) {
        total += load_agent(partials + slot);
    }
    __syncthreads();  // scratch is reused
    total = block_reduce_sum<Waves>(total, scratch);
    if (threadIdx.x == 0) {
        *res = static_cast<ResType>(total);
    }
    // re-arm every counter for the next call on this stream (all arrivals of
    // this launch have happened: this workgroup drew the last ticket)
    if (threadIdx.x <= dot_ticket_shards) {
        __hip_atomic_store(ticket + 16 * threadIdx.x, 0u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}


}  // namespace kernel


/** Device scratch of a handle that the dot launcher needs. */
struct dot_workspace {
    void *partials;          // dot_max_partials * 8 bytes
    unsigned int *ticket;    // dot_ticket_words words, zero between calls
    void *scalar;            // 16-byte slot (`get_device_value_ptr`)
    int num_cus;
};


template <typename ArType, typename Config, typename ResType, typename XView,
          typename YView>
hipError_t launch_dot_with(const dot_workspace &ws, bool atomic,
                           const XView &x, const YView &y, ResType *res,
                           hipStream_t stream)
{
    using x_traits = view_traits<XView>;
    using y_traits = view_traits<YView>;
    using st_type = std::remove_cv_t<typename x_traits::storage_type>;
    constexpr int vec16 = 16 / sizeof(st_type);
    constexpr int waves = Config::waves;

    const index_type n = x.length(0);
    int blocks = ws.num_cus * Config::blocks_per_cu;
    if (blocks > dot_max_partials) {
        blocks = dot_max_partials;
    }
    // every workgroup should have at least two full unrolled iterations of
    // work; smaller problems get a smaller grid (fewer arrivals to combine)
    const index_type per_block =
        index_type{waves} * wave_size * vec16 * Config::unroll * 2;
    const index_type needed = n > 0 ? (n + per_block - 1) / per_block : 1;
    if (needed < blocks) {
        blocks = static_cast<int>(needed);
    }
    const dim3 grid(blocks, 1, 1);
    const dim3 block(waves * wave_size, 1, 1);
    // the 16-byte path reads both operands as plain storage of ONE type;
    // other accessor kinds and mixed storage types go through operator()
    constexpr bool can_vec =
        x_traits::is_pointer_view && y_traits::is_pointer_view &&
        std::is_same<st_type, std::remove_cv_t<
                                  typename y_traits::storage_type>>::value;
    constexpr int vec = can_vec ? vec16 : 1;
    const bool vec_ok = can_vec && (x_traits::stride(x) == 1 || n <= 1) &&
                        (y_traits::stride(y) == 1 || n <= 1) &&
                        is_aligned(x_traits::data(x), 16) &&
                        is_aligned(y_traits::data(y), 16);
    auto *partials = static_cast<ArType *>(ws.partials);

    if (atomic) {
        // reference sequence: init_res -> dot (atomic_add) -> cast_result
        constexpr bool direct = std::is_same<ArType, ResType>::value;
        ArType *accum = direct ? reinterpret_cast<ArType *>(res)
                               : static_cast<ArType *>(ws.scalar);
        kernel::init_res<<<1, 64, 0, stream>>>(accum);
        if (vec_ok) {
            kernel::dot<Config, vec, true>
                <<<grid, block, 0, stream>>>(x, y, accum, ws.ticket, res);
        } else {
            kernel::dot<Config, 1, true>
                <<<grid, block, 0, stream>>>(x, y, accum, ws.ticket, res);
        }
        if (!direct) {
            kernel::cast_result<<<1, 64, 0, stream>>>(accum, res);
        }
    } else {
        if (vec_ok) {
            kernel::dot<Config, vec, false>
                <<<grid, block, 0, stream>>>(x, y, partials, ws.ticket, res);
        } else {
            kernel::dot<Config, 1, false>
                <<<grid, block, 0, stream>>>(x, y, partials, ws.ticket, res);
        }
    }
    return hipGetLastError();
}


/** Launches with an explicit Config, or (default) picks the production shape
 *  from the amount of data streamed. */
template <typename ArType, typename Config = dot_auto_config, typename ResType,
          typename XView, typename YView>
hipError_t launch_dot(const dot_workspace &ws, bool atomic, const XView &x,
                      const YView &y, ResType *res, hipStream_t stream)
{
    if (!std::is_same<Config, dot_auto_config>::value) {
        using chosen = std::conditional_t<
            std::is_same<Config, dot_auto_config>::value, dot_stride_config,
            Config>;
        return launch_dot_with<ArType, chosen>(ws, atomic, x, y, res, stream);
    }
    using st_type = typename view_traits<XView>::storage_type;
    const index_type bytes =
        2 * static_cast<index_type>(sizeof(st_type)) * x.length(0);
    if (bytes >= dot_contiguous_min_bytes) {
        return launch_dot_with<ArType, dot_contiguous_config>(ws, atomic, x, y,
                                                              res, stream);
    }
    return launch_dot_with<ArType, dot_stride_config>(ws, atomic, x, y, res,
                                                      stream);
}


}  // namespace dev
}  // namespace accblas
