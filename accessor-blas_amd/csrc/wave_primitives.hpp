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

/** A pointer into global memory, typed as one. Pointer parameters of a KERNEL
 *  are known to point there; a `T *` that reaches an out-of-line device
 *  function as an argument is a flat pointer and its loads are `flat_load`s,
 *  which count in vmcnt AND lgkmcnt: every wait for an LDS read then waits for
 *  every matrix load in flight. (A cast to this type and back inside the
 *  function is folded away before the compiler's address-space inference sees
 *  it, and `__builtin_assume(!is_shared && !is_private)` is not picked up: the
 *  PARAMETER has to have this type.) */
template <typename T>
using global_ptr = T __attribute__((address_space(1))) *;

/** `value` (the same in every lane) as scalar registers: lane 0's copy, word
 *  by word. Arguments of an out-of-line device function arrive in vector
 *  registers whatever they hold; what is computed from them would stay there. */
template <typename T>
__device__ __forceinline__ T wave_uniform(const T &value)
{
    static_assert(sizeof(T) % 4 == 0 && std::is_trivially_copyable<T>::value,
                  "whole 32-bit words");
    std::uint32_t words[sizeof(T) / 4];
    __builtin_memcpy(words, &value, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) {
        words[i] = __builtin_amdgcn_readfirstlane(words[i]);
    }
    T out = value;  // (T need not be default-constructible)
    __builtin_memcpy(&out, words, sizeof(T));
    return out;
}


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


/** DPP controls used below (data-parallel-primitive lane moves run on the
 *  vector ALU: a few cycles, where a `ds_bpermute` shuffle costs a trip through
 *  the LDS crossbar). */
constexpr int dpp_quad_swap_1 = 0xB1;   // quad_perm:[1,0,3,2]: lane ^ 1
constexpr int dpp_quad_swap_2 = 0x4E;   // quad_perm:[2,3,0,1]: lane ^ 2
constexpr int dpp_row_rotate_4 = 0x124;  // row_ror:4 inside each 16-lane row
constexpr int dpp_row_rotate_8 = 0x128;  // row_ror:8
constexpr int dpp_row_rotate_12 = 0x12C;  // row_ror:12: lane l reads lane l + 4 (mod 16)

/** `value` of the lane that DPP control `Ctrl` names (4- and 8-byte types). */
template <int Ctrl, typename T>
__device__ __forceinline__ T dpp_move(T value)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte values only");
    int words[sizeof(T) / 4];
    __builtin_memcpy(words, &value, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) {
        words[i] = __builtin_amdgcn_update_dpp(0, words[i], Ctrl, 0xF, 0xF, true);
    }
    T out = value;  // (T need not be default-constructible)
    __builtin_memcpy(&out, words, sizeof(T));
    return out;
}


/** Both halves of a lane-bit-4 / lane-bit-5 exchange in one vector-ALU
 *  instruction per 32-bit word (gfx950: `v_permlane16_swap_b32`,
 *  `v_permlane32_swap_b32`), where `__shfl_xor(v, 16 | 32)` is a
 *  `ds_bpermute_b32` through the LDS crossbar. With the same register as both
 *  operands the swap leaves `low` = the value of the lane whose bit is CLEAR
 *  and `high` = the value of the lane whose bit is SET, in both lanes of every
 *  pair. */
template <typename T>
struct lane_pair {
    T low;
    T high;
};

template <int Bit, typename T>
__device__ __forceinline__ lane_pair<T> lane_bit_pair(T value)
{
    static_assert(Bit == 4 || Bit == 5, "lane bit 4 (rows of 16) or 5 (halves)");
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte values only");
    unsigned int words[sizeof(T) / 4];
    unsigned int low[sizeof(T) / 4];
    unsigned int high[sizeof(T) / 4];
    __builtin_memcpy(words, &value, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) {
        const auto both =
            Bit == 4 ? __builtin_amdgcn_permlane16_swap(words[i], words[i],
                                                        false, false)
                     : __builtin_amdgcn_permlane32_swap(words[i], words[i],
                                                        false, false);
        low[i] = both[0];
        high[i] = both[1];
    }
    lane_pair<T> out;
    __builtin_memcpy(&out.low, low, sizeof(T));
    __builtin_memcpy(&out.high, high, sizeof(T));
    return out;
}

/** `value` of the lane whose lane bit `Bit` (4 or 5) differs:
 *  `__shfl_xor(value, 1 << Bit)` without the LDS crossbar. */
template <int Bit, typename T>
__device__ __forceinline__ T lane_bit_partner(T value, int lane)
{
    const lane_pair<T> both = lane_bit_pair<Bit>(value);
    return ((lane >> Bit) & 1) ? both.low : both.high;
}

/** value + `__shfl_xor(value, 1 << Bit)`, same bits in both lanes of a pair
 *  (the sum of the same two numbers in either order). */
template <int Bit, typename T>
__device__ __forceinline__ T lane_bit_pair_sum(T value)
{
    const lane_pair<T> both = lane_bit_pair<Bit>(value);
    return both.low + both.high;
}


/**
 * Sums Count values per lane over groups of `Width` consecutive lanes
 * (Count a power of two <= 16, Width = 8, 16, 32 or 64) in
 * log2(Width) exchange steps that HALVE the number of values a lane carries
 * (a transposing butterfly): Count - 1 + log2(Width / Count) lane moves per
 * lane instead of Count * log2(Width). Exchanges over lane bits 0..3 are DPP
 * moves (bits 2 and 3 as rotations inside the 16-lane row, which pair every
 * lane with one whose bit differs), bits 4 and 5 `lane_bit_partner` (one
 * `v_permlane16/32_swap` per word). Every lane of a
 * group ends with a total of value index `lane & (Count - 1)`; fixed order,
 * so the result is reproducible. ONLY lanes [0, Count) of each group hold the
 * canonical total: the rotations used for lane bits 2 and 3 are one-way moves,
 * not swaps, so when such a step is a plain all-reduce (Count <= 4 or 8) the
 * lanes of different quads add the same partial sums in different association
 * orders and may differ from each other in the last bit. Callers read lanes
 * [0, Count) (`col_group < NI` in the TRSV kernels). Width = 8: the rotation
 * over lane bit 2 reaches into the neighbouring group for lanes 4..7 of a
 * group, so there lanes [0, Count) are the only ones with a total at all.
 */
template <int Count, int Width, typename T>
__device__ __forceinline__ T transpose_reduce_sum(const T (&values)[Count],
                                                  int lane)
{
    static_assert(Count >= 1 && (Count & (Count - 1)) == 0 && Count <= 16,
                  "Count must be a power of two <= 16");
    static_assert((Width == 8 || Width == 16 || Width == 32 || Width == 64) &&
                      Count <= Width && (Width > 8 || Count <= 4),
                  "Width must be 8 (Count <= 4), 16, 32 or 64 lanes");
    T cur[Count];
#pragma unroll
    for (int i = 0; i < Count; ++i) {
        cur[i] = values[i];
    }
    // from[bit](v) = v of the lane paired with this one across lane bit `bit`
    const auto from = [lane](auto bit_tag, T v) -> T {
        constexpr int bit = decltype(bit_tag)::value;
        if (bit == 0) {
            return dpp_move<dpp_quad_swap_1>(v);
        } else if (bit == 1) {
            return dpp_move<dpp_quad_swap_2>(v);
        } else if (bit == 2) {
            // (row_ror:4 hands lane l the value of lane l - 4: any partner
            // inside a 16-lane group will do, in a group of 8 lanes 0..3 need
            // lanes 4..7)
            return Width == 8 ? dpp_move<dpp_row_rotate_12>(v)
                              : dpp_move<dpp_row_rotate_4>(v);
        } else if (bit == 3) {
            return dpp_move<dpp_row_rotate_8>(v);
        } else {
            return lane_bit_partner<(bit < 4 ? 4 : bit)>(v, lane);
        }
    };
    constexpr int halving_steps = Count == 1 ? 0 : (Count == 2 ? 1 : (Count == 4 ? 2 : (Count == 8 ? 3 : 4)));
    constexpr int all_steps =
        Width == 8 ? 3 : (Width == 16 ? 4 : (Width == 32 ? 5 : 6));
    // halving steps: a lane keeps the values whose index bit equals its own
    // lane bit and receives the partner's copies of those
    const auto halve = [&](auto bit_tag, int live) {
        constexpr int bit = decltype(bit_tag)::value;
        const bool mine = (lane >> bit) & 1;
#pragma unroll
        for (int j = 0; j < Count / 2; ++j) {
            if (j < live / 2) {
                const T keep = mine ? cur[2 * j + 1] : cur[2 * j];
                const T send = mine ? cur[2 * j] : cur[2 * j + 1];
                cur[j] = keep + from(bit_tag, send);
            }
        }
    };
    if (halving_steps > 0) halve(std::integral_constant<int, 0>{}, Count);
    if (halving_steps > 1) halve(std::integral_constant<int, 1>{}, Count / 2);
    if (halving_steps > 2) halve(std::integral_constant<int, 2>{}, Count / 4);
    if (halving_steps > 3) halve(std::integral_constant<int, 3>{}, Count / 8);
    T total = cur[0];
    // the remaining lane bits: plain all-reduce of the one value left
    if (halving_steps <= 0 && all_steps > 0) total += from(std::integral_constant<int, 0>{}, total);
    if (halving_steps <= 1 && all_steps > 1) total += from(std::integral_constant<int, 1>{}, total);
    if (halving_steps <= 2 && all_steps > 2) total += from(std::integral_constant<int, 2>{}, total);
    if (halving_steps <= 3 && all_steps > 3) total += from(std::integral_constant<int, 3>{}, total);
    if (all_steps > 4) total += from(std::integral_constant<int, 4>{}, total);
    if (all_steps > 5) total += from(std::integral_constant<int, 5>{}, total);
    return total;
}


/**
 * Sum over the 64 lanes of a wave; the total is valid in lane 0 (other lanes
 * hold partial garbage). The tree of `__shfl_down` with offsets 32, 16, 8, 4,
 * 2, 1 - lane 0 ends with exactly those bits - on vector-ALU lane moves
 * (`v_permlane32/16_swap`, DPP row rotations and quad permutations) instead of
 * six trips through the LDS crossbar per value: a GEMV workgroup reduces up to
 * four rows this way at the end of its short life.
 */
template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T value)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4/8-byte values only");
    value = lane_bit_pair_sum<5>(value);               // l + (l + 32)
    value = lane_bit_pair_sum<4>(value);               // + (l + 16)
    value += dpp_move<dpp_row_rotate_8>(value);        // + (l + 8) for l < 8
    value += dpp_move<dpp_row_rotate_12>(value);       // + (l + 4) for l < 4
    value += dpp_move<dpp_quad_swap_2>(value);         // + (l + 2) for l < 2
    value += dpp_move<dpp_quad_swap_1>(value);         // lane 0 + lane 1
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

/** Workgroup barrier that orders LDS traffic only: waits for this wave's LDS
 *  operations, then `s_barrier`. `__syncthreads()` also drains the wave's
 *  outstanding GLOBAL loads (`s_waitcnt vmcnt(0)`), which stalls a software
 *  pipeline that keeps prefetches in flight across the barrier. Use it where
 *  the data handed over lives in LDS and nothing read from global memory is
 *  published by the barrier. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/** Release / acquire between the waves of a workgroup for data that lives in
 *  LDS: a wave's LDS operations are served in order, so "every earlier LDS
 *  operation of this wave has been performed" (release) is one wait on the LDS
 *  counter and "later LDS reads see what the writer wrote before the flag"
 *  (acquire) needs no instruction at all - both only have to keep the COMPILER
 *  from moving LDS accesses across them. `__builtin_amdgcn_fence(...,
 *  "workgroup")` also waits for the wave's outstanding GLOBAL loads
 *  (`s_waitcnt vmcnt(0)`): in a sweep that keeps matrix prefetches in flight
 *  that is one full memory latency per step. */
__device__ __forceinline__ void lds_release()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void lds_acquire()
{
    asm volatile("" ::: "memory");
}

/** A 32-bit flag in LDS that waves of a workgroup poll / raise. Relaxed
 *  workgroup-scope atomics compile to `ds_read_b32` / `ds_write_b32`; a
 *  `volatile __shared__` object reached through a reference (a lambda capture)
 *  is accessed with FLAT loads instead - the address-space inference leaves
 *  volatile accesses alone -, and a flat load counts in `vmcnt`: every look at
 *  the flag then waits for all the global loads the wave has in flight. */
__device__ __forceinline__ std::int32_t lds_flag_load(const std::int32_t *flag)
{
    return __hip_atomic_load(flag, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void lds_flag_store(std::int32_t *flag,
                                               std::int32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
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
