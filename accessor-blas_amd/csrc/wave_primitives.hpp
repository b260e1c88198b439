// gfx950 wave64 primitives: the MI355X counterpart of the reference's
// `cuda/kernel_utils.cuh` (sub-warp butterfly `reduce` :30-53, block `reduce`
// :69-89, WARP_SIZE=32 :14) and `cuda/atomics.cuh` (`atomic_add` :80-103).
//
// Not a translation: a CDNA4 wavefront has 64 lanes, so reductions are 6-step
// 64-lane shuffles and a workgroup combines one value per WAVE through LDS
// (at most 16 values) instead of the reference's log2(block) barrier tree.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>


namespace accblas {
namespace dev {


constexpr int wave_size = 64;


/** Lane index inside the wave (0..63) for a 1-D workgroup. */
__device__ __forceinline__ int lane_id() { return threadIdx.x & (wave_size - 1); }

/** Wave index inside a 1-D workgroup. */
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }


/**
 * All-reduce over `width` consecutive lanes (power of two, <= 64) with a
 * butterfly: every lane of the group ends with the same value. Order of the
 * partner exchange: xor 1, 2, 4, ... (as `kernel_utils.cuh:46-51`, extended to
 * 64 lanes).
 */
template <int width = wave_size, typename T, typename Op>
__device__ __forceinline__ T group_all_reduce(T value, Op op)
{
    static_assert(width > 0 && width <= wave_size && (width & (width - 1)) == 0,
                  "width must be a power of two <= 64");
#pragma unroll
    for (int mask = 1; mask < width; mask <<= 1) {
        const T other = __shfl_xor(value, mask, width);
        value = op(value, other);
    }
    return value;
}


/**
 * Sum over the 64 lanes of a wave; the total is valid in lane 0 (other lanes
 * hold partial garbage). Uses shfl_down: offsets 32, 16, 8, 4, 2, 1.
 */
template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T value)
{
#pragma unroll
    for (int offset = wave_size / 2; offset > 0; offset >>= 1) {
        value += __shfl_down(value, offset, wave_size);
    }
    return value;
}


/**
 * Sum over a 1-D workgroup of `num_waves` waves. `scratch` must hold
 * `num_waves` values in LDS. The total is returned to thread 0 only (like the
 * reference's block reduce, whose result only thread 0 may read,
 * `kernel_utils.cuh:83-88`). Fixed order: waves are added 0, 1, 2, ... so the
 * result is reproducible for a given workgroup shape.
 */
template <int num_waves, typename T>
__device__ __forceinline__ T block_reduce_sum(T value, T *scratch)
{
    value = wave_reduce_sum(value);
    if (num_waves == 1) {
        return value;
    }
    if (lane_id() == 0) {
        scratch[wave_id()] = value;
    }
    __syncthreads();
    T total{};
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < num_waves; ++w) {
            total += scratch[w];
        }
    }
    return total;
}


namespace detail {


template <typename T, typename = void>
struct atomic_helper {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8,
                  "atomic_add needs a 4- or 8-byte type");
};

// Compare-and-swap fallback for 4- and 8-byte types without a native add
// (`atomics.cuh:36-73` does the same on `atomicCAS`).
template <typename T>
struct atomic_helper<T, std::enable_if_t<sizeof(T) == 4 || sizeof(T) == 8>> {
    using bits_type =
        std::conditional_t<sizeof(T) == 4, unsigned int, unsigned long long>;

    __device__ __forceinline__ static T add(T *addr, T value)
    {
        auto *bits_addr = reinterpret_cast<bits_type *>(addr);
        bits_type expected = __hip_atomic_load(bits_addr, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
        bits_type desired;
        do {
            T current;
            __builtin_memcpy(&current, &expected, sizeof(T));
            const T updated = current + value;
            __builtin_memcpy(&desired, &updated, sizeof(T));
        } while (!__hip_atomic_compare_exchange_strong(
            bits_addr, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT));
        T old;
        __builtin_memcpy(&old, &expected, sizeof(T));
        return old;
    }
};


}  // namespace detail


/** Generic atomic add (agent scope); returns the previous value. */
template <typename T>
__device__ __forceinline__ T atomic_add(T *addr, T value)
{
    return detail::atomic_helper<T>::add(addr, value);
}

// Types with a native gfx950 read-modify-write (`atomics.cuh:88-110`); float
// and double compile to single global_atomic_add_f32 / _f64 instructions.
#define ACCBLAS_NATIVE_ATOMIC_ADD(_type)                                  \
    __device__ __forceinline__ _type atomic_add(_type *addr, _type value) \
    {                                                                     \
        return atomicAdd(addr, value);                                    \
    }
ACCBLAS_NATIVE_ATOMIC_ADD(int)
ACCBLAS_NATIVE_ATOMIC_ADD(unsigned int)
ACCBLAS_NATIVE_ATOMIC_ADD(unsigned long long)
ACCBLAS_NATIVE_ATOMIC_ADD(float)
ACCBLAS_NATIVE_ATOMIC_ADD(double)
#undef ACCBLAS_NATIVE_ATOMIC_ADD


/** Relaxed agent-scope (sc1, L1-bypassing) store / load of one 4/8-byte word:
 *  the payload form of the inter-workgroup hand-off (guide G16, recipe R1). */
template <typename T>
__device__ __forceinline__ void store_agent(T *addr, T value)
{
    using bits_type =
        std::conditional_t<sizeof(T) == 4, unsigned int, unsigned long long>;
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte words only");
    bits_type bits;
    __builtin_memcpy(&bits, &value, sizeof(T));
    __hip_atomic_store(reinterpret_cast<bits_type *>(addr), bits,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__device__ __forceinline__ T load_agent(const T *addr)
{
    using bits_type =
        std::conditional_t<sizeof(T) == 4, unsigned int, unsigned long long>;
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte words only");
    const bits_type bits = __hip_atomic_load(
        reinterpret_cast<const bits_type *>(addr), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    T value;
    __builtin_memcpy(&value, &bits, sizeof(T));
    return value;
}

/** Drains this wave's outstanding vector-memory operations. Written as asm
 *  so the compiler cannot drop it in front of a flag/ticket update (guide:
 *  'Compiler hazard'). */
__device__ __forceinline__ void drain_vmem()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


}  // namespace dev
}  // namespace accblas
