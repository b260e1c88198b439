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
//     workgroup hands its partial over as a TAGGED slot ("the data is the
//     flag": {launch serial, value bits} written with one write-through store,
//     nothing to drain, no ticket); the last workgroup of the grid collects:
//     it polls the slots, adds the partials in slot order, writes the result
//     already cast to ResType (so `init_res` and `cast_result` disappear) and
//     clears the tags it consumed;
//   * there is no other state: a slot whose tag is not this launch's serial
//     (never written, left by an aborted earlier launch, scribbled over) is
//     simply "not there yet", so nothing has to be re-armed between calls and
//     nothing a previous call left behind can poison this one. The hand-off
//     does not depend on dispatch order or XCD placement; only the collector
//     waits, and only for workgroups that need nothing from it.
// ACCBLAS_DOT_ATOMIC keeps the reference's behaviour (atomic_add per
// workgroup) for comparison.
#pragma once

#include <hip/hip_runtime.h>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


constexpr int dot_max_partials = 4096;    // workspace slots in the handle
using dot_slot_word = unsigned long long;
constexpr int dot_slot_words = 2;         // {tag, low 32 bits}, {tag, high 32 bits}
constexpr std::size_t dot_slot_bytes =
    std::size_t{dot_max_partials} * dot_slot_words * sizeof(dot_slot_word);
// The collector gives up after this many unsuccessful polls of one slot
// (seconds; a healthy launch needs microseconds) and returns NaN.
#ifndef ACCBLAS_DOT_SPIN_LIMIT
#define ACCBLAS_DOT_SPIN_LIMIT (1u << 22)
#endif
constexpr unsigned int dot_spin_limit = ACCBLAS_DOT_SPIN_LIMIT;
// what the collector leaves in `dot_workspace::status` when it gave up
constexpr std::uint32_t dot_expired_word = 0xdeadbeefu;


namespace detail {


/** One partial sum as one (4-byte types) or two (8-byte types) 8-byte words
 *  {tag : 32 | payload : 32}. Every word is written with a single agent-scope
 *  (write-through) store, so a reader sees either the old word or the new one,
 *  never a mix: the tag validates the payload next to it. */
template <typename T>
struct dot_slot {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte partials only");
    static constexpr int words = sizeof(T) / 4;

    __device__ __forceinline__ static void publish(dot_slot_word *slots,
                                                   unsigned int slot, T value,
                                                   std::uint32_t tag)
    {
        const dot_slot_word high = dot_slot_word{tag} << 32;
        dot_slot_word bits = 0;
        __builtin_memcpy(&bits, &value, sizeof(T));
        __hip_atomic_store(slots + dot_slot_words * slot,
                           high | (bits & 0xffffffffull), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (words == 2) {
            __hip_atomic_store(slots + dot_slot_words * slot + 1,
                               high | (bits >> 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    /** True when the slot carries `tag`; `value` is then its payload. */
    __device__ __forceinline__ static bool fetch(const dot_slot_word *slots,
                                                 unsigned int slot,
                                                 std::uint32_t tag, T &value)
    {
        dot_slot_word raw[words];
#pragma unroll
        for (int w = 0; w < words; ++w) {
            raw[w] = __hip_atomic_load(slots + dot_slot_words * slot + w,
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
#pragma unroll
        for (int w = 0; w < words; ++w) {
            ok = ok && (raw[w] >> 32) == tag;
        }
        const dot_slot_word bits =
            (raw[0] & 0xffffffffull) | (words == 2 ? raw[words - 1] << 32 : 0);
        __builtin_memcpy(&value, &bits, sizeof(T));
        return ok;
    }

    /** Clears the tag (a later launch with the same serial - a replayed
     *  hipGraph - must not take this launch's value for its own). */
    __device__ __forceinline__ static void clear(dot_slot_word *slots,
                                                 unsigned int slot)
    {
#pragma unroll
        for (int w = 0; w < words; ++w) {
            __hip_atomic_store(slots + dot_slot_words * slot + w,
                               dot_slot_word{0}, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};


}  // namespace detail


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
 *  tools/tune_dot_chunk.hip; profiles/r01_tune_dot.txt, r02_tune_dot*.txt,
 *  r03_tune_dot_chunk.txt): 512-thread workgroups, ONE per CU. Long streams run
 *  fastest when each workgroup sweeps one contiguous slice with 8 load pairs
 *  per lane in flight (256 KB per CU; 2 x 1 GiB fp32: 306 vs 310 us with two
 *  workgroups of 4 pairs, 1 GiB of fp16: 156 us against 177 us for a strided
 *  grid); shorter ones when the whole grid strides together with 4 pairs. The
 *  two cross between 256 and 512 MiB of data (profiles/r02_tune_dot_rule.txt).
 *  Until round 3 the strided grid had two workgroups per CU; one is as fast
 *  below 32 MiB and 1-7 % faster from there to 512 MiB with the data coming
 *  from HBM (rotating buffers, us per launch, two / one per CU: 64 MiB 13.85 /
 *  12.97, 128 MiB - the per-GPU share of config C4 on 8 GPUs - 22.26 / 22.15
 *  on one box and 24.87 / 22.13 on another, 256 MiB 41.6 / 40.7, 512 MiB 81.2
 *  / 78.6), and the collector has half as many slots to gather. */
using dot_stride_config = dot_config<8, 4, 1, false>;
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
 * Grid-stride dot product. Vec > 1 requires unit increments and plain storage
 * of one type (checked by the launcher; any base address); Vec == 1 goes
 * through operator() only and works for any increment.
 *
 * Atomic == false: tagged slot per workgroup, collected by the last workgroup
 * of the grid in slot order and written as ResType; `serial` != 0 is the tag
 * of this launch. Atomic == true: atomic_add(accum, partial) as
 * `cuda/dot_kernels.cuh:158-160` (accum must be zeroed before).
 */
template <typename Config, int Vec, bool Atomic, typename XView,
          typename YView, typename ArType, typename ResType>
__global__ __launch_bounds__(Config::waves *wave_size) void dot(
    XView x, YView y, ArType *accum, dot_slot_word *slots,
    std::uint32_t serial, ResType *res, std::uint32_t *status)
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
        // slices start a multiple of 4 KiB (256 chunks of 16 bytes) behind the
        // operands' bases - on 4 KiB boundaries when the bases are
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
    __shared__ ar_type own_sum;
    const ar_type block_sum = block_reduce_sum<Waves>(local, scratch);

    if (Atomic) {
        if (threadIdx.x == 0) {
            atomic_add(accum, block_sum);
        }
        return;
    }

    // Hand-off: every workgroup but the last of the grid publishes its partial
    // as a tagged slot and is done (nothing to drain, no ticket: the tag
    // arrives with the value). The last workgroup collects.
    using slot_io = detail::dot_slot<ar_type>;
    const bool collector = blockIdx.x == gridDim.x - 1;
    if (threadIdx.x == 0) {
        if (collector) {
            own_sum = block_sum;
        } else {
            slot_io::publish(slots, blockIdx.x, block_sum, serial);
        }
    }
    if (!collector) {
        return;
    }
    __syncthreads();  // own_sum written, scratch free again
    // fixed-order sum of the partials: thread t adds slots t, t + threads, ...
    ar_type total{};
    bool expired = false;
    for (unsigned int slot = threadIdx.x; slot < gridDim.x;
         slot += block_threads) {
        ar_type value = own_sum;
        if (slot != blockIdx.x) {
            unsigned int spins = 0;
            while (!slot_io::fetch(slots, slot, serial, value)) {
                if (++spins >= dot_spin_limit) {
                    expired = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            slot_io::clear(slots, slot);
        }
        total += value;
    }
    total = block_reduce_sum<Waves>(total, scratch);
    const bool any_expired = __syncthreads_or(expired);
    if (threadIdx.x == 0) {
        // a wait that gave up (the other workgroups of the grid did not get
        // a CU for seconds, e.g. behind another stream's kernel) must not
        // look like a result: NaN, and a sticky word the host can ask for
        // (`accblas_dot_status`)
        *res = any_expired ? static_cast<ResType>(__builtin_nanf(""))
                           : static_cast<ResType>(total);
        if (any_expired && status != nullptr) {
            __hip_atomic_store(status, dot_expired_word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


}  // namespace kernel


/** Device scratch of a handle that the dot launcher needs. */
struct dot_workspace {
    dot_slot_word *slots;    // dot_slot_bytes: one tagged slot per workgroup
    void *scalar;            // 16-byte slot (`get_device_value_ptr`)
    std::uint32_t *status = nullptr;  // sticky "a collector gave up" word, or NULL
    int num_cus;
    std::uint32_t serial;    // tag of the NEXT launch (host side, never 0)

    /** Tag for one launch; launches that share the slots must not share a
     *  tag unless one has completed before the other starts (stream order). */
    std::uint32_t next_serial()
    {
        const std::uint32_t tag = serial;
        serial = serial + 1 == 0 ? 1 : serial + 1;
        return tag;
    }
};


template <typename ArType, typename Config, typename ResType, typename XView,
          typename YView>
hipError_t launch_dot_with(dot_workspace &ws, bool atomic,
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
    // (any base: the 16-byte loads need no alignment, `load_packed`; a
    // subvector that starts off a 16-byte boundary ran element-wise until
    // round 3, profiles/r04_dot_unaligned.txt)
    const bool vec_ok = can_vec && (x_traits::stride(x) == 1 || n <= 1) &&
                        (y_traits::stride(y) == 1 || n <= 1) &&
                        is_aligned(x_traits::data(x), sizeof(st_type)) &&
                        is_aligned(y_traits::data(y), sizeof(st_type));
    if (atomic) {
        // reference sequence: init_res -> dot (atomic_add) -> cast_result
        constexpr bool direct = std::is_same<ArType, ResType>::value;
        ArType *accum = direct ? reinterpret_cast<ArType *>(res)
                               : static_cast<ArType *>(ws.scalar);
        kernel::init_res<<<1, 64, 0, stream>>>(accum);
        if (vec_ok) {
            kernel::dot<Config, vec, true>
                <<<grid, block, 0, stream>>>(x, y, accum, ws.slots, 0u, res,
                                             ws.status);
        } else {
            kernel::dot<Config, 1, true>
                <<<grid, block, 0, stream>>>(x, y, accum, ws.slots, 0u, res,
                                             ws.status);
        }
        if (!direct) {
            kernel::cast_result<<<1, 64, 0, stream>>>(accum, res);
        }
    } else {
        ArType *const unused = nullptr;
        const std::uint32_t serial = ws.next_serial();
        if (vec_ok) {
            kernel::dot<Config, vec, false>
                <<<grid, block, 0, stream>>>(x, y, unused, ws.slots, serial, res,
                                             ws.status);
        } else {
            kernel::dot<Config, 1, false>
                <<<grid, block, 0, stream>>>(x, y, unused, ws.slots, serial, res,
                                             ws.status);
        }
    }
    return hipGetLastError();
}


/** Launches with an explicit Config, or (default) picks the production shape
 *  from the amount of data streamed. */
template <typename ArType, typename Config = dot_auto_config, typename ResType,
          typename XView, typename YView>
hipError_t launch_dot(dot_workspace &ws, bool atomic, const XView &x,
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
