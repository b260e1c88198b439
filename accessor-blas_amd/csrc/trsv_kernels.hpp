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
//     write-through (sc1) store; consumers poll the granules they need with
//     sc1 loads. No progress word, no drain, no fence on the critical path
//     (the reference's __threadfence + volatile spin,
//     `trsv_kernels.cuh:171-182,229-233`, would be 5 serialised memory round
//     trips per hop here); per-XCD L2s are not coherent and L1 is never
//     refreshed, so plain loads/stores are not an option;
//   * ONE launch per solve: the tag is the solve's epoch, read from a small
//     device-resident state that the last workgroup to finish re-arms for the
//     next solve (`trsv_state`), so nothing is zeroed between solves, the
//     reference's `trsv_init` launch disappears and a captured solve can be
//     replayed from a hipGraph. A block's 64 granules are 512-byte aligned
//     (4 cache lines; a 64-byte misalignment cost 4-8 % of the whole solve);
//   * ONE workgroup per CU (`trsv_single_block_pad`): a second resident block
//     would sweep on the SIMDs of the block that is on the chain;
//   * per block: all waves stage the diagonal tile AND the last off-diagonal
//     panel in LDS; the tile is inverted recursively (four 16 x 16 forward
//     substitutions by the four 16-lane groups of wave 0, lanes owning
//     columns in registers, then two merge levels X10 = -X11 C X00 as MFMA
//     products) instead of a Gauss-Jordan sweep through shared memory with a
//     sync per step;
//   * far panels (all but the last): 16-byte loads of the storage type,
//     four panels deep in registers together with the x granules of those
//     panels, FMA in the arithmetic type, ONE cross-lane reduction at the end;
//   * the last panel is folded with the inverse ahead of time on the matrix
//     cores (16 x 16 x 4 MFMA in the arithmetic type),
//     M = inv(T_bb) A_{b,b-1}, so that x_b = inv(T_bb) rhs_far - M x_{b-1}:
//     after x_{b-1} arrives ONE 64 x 64 mat-vec from LDS is left (lane = row,
//     columns split over the waves, vector entries broadcast with v_readlane,
//     per-wave partials combined through LDS, one barrier).
#pragma once

// (matrix loads of the TRSV kernels: non-temporal unless a build says otherwise)
#ifndef ACCBLAS_TRSV_NT
#define ACCBLAS_TRSV_NT true
#endif

#include <hip/hip_runtime.h>

#include <atomic>

#include "views.hpp"
#include "wave_primitives.hpp"


namespace accblas {
namespace dev {


constexpr int trsv_tile = wave_size;  // 64
constexpr int trsv_waves = 8;         // waves per workgroup
#ifndef ACCBLAS_TRSV_PREFETCH_DEPTH
#define ACCBLAS_TRSV_PREFETCH_DEPTH 4
#endif
constexpr int trsv_prefetch_depth =
    ACCBLAS_TRSV_PREFETCH_DEPTH;  // far panels in flight per wave

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
// (the stamps live in wave 0's registers and are written once, at the end of
// the block: a store - or the load of this pointer - inside the sweep would
// sit in the memory queues the stamps are meant to observe)
__device__ unsigned long long *trsv_diag_buffer;
#define ACCBLAS_TRSV_STAMP(_block, _slot) diag_stamp[_slot] = wall_clock64()
#define ACCBLAS_TRSV_CYCLES(_block, _slot) diag_stamp[_slot] = clock64()
// first time the far sweep of wave 0 had to wait for a panel's x: when, and
// which panel (slots 13, 14)
#define ACCBLAS_TRSV_FIRST_WAIT(_panel)                                     \
    do {                                                                    \
        if (diag_stamp[13] == 0) {                                          \
            diag_stamp[13] = wall_clock64();                                \
            diag_stamp[14] = static_cast<unsigned long long>(_panel);       \
        }                                                                   \
    } while (false)
#else
#define ACCBLAS_TRSV_STAMP(_block, _slot) \
    do {                                  \
    } while (false)
#define ACCBLAS_TRSV_CYCLES(_block, _slot) \
    do {                                   \
    } while (false)
#define ACCBLAS_TRSV_FIRST_WAIT(_panel) \
    do {                                \
    } while (false)
#endif


/**
 * Device-resident state of the TRSV hand-off workspace (its header). The
 * workspace is zeroed and this state initialised once, when the buffer is
 * allocated; after that every solve is ONE launch that re-arms the workspace
 * for its successor (so a captured launch can be replayed from a hipGraph):
 *   - granules of a solve carry `epoch` in their tag word (0 = never written,
 *     1 = unused, >= 2 = epochs);
 *   - row blocks are numbered by tickets drawn from `tickets`, a 64-bit word
 *     {key : 32 | next ticket : 32}. Between solves it is {trsv_clean_key, 0}.
 *     The workgroup that draws ticket 0 of a clean word adds its launch's
 *     serial number (a kernel argument, different for every launch the host
 *     enqueues) to the key, so the word reads {serial, t} while the launch
 *     runs; both keys are good for the launch's own workgroups;
 *   - the last workgroup to finish (counted in `finished`) puts the word back
 *     to {trsv_clean_key, 0}, resets the other counters and moves `epoch` on;
 *     when the epoch would wrap it zeroes the granules and starts again at 2;
 *   - a word that carries any OTHER key - the serial of a launch that never
 *     finished, or bytes somebody scribbled over the header - is recognised by
 *     the first workgroups of the next launch: one of them (a compare-and-swap
 *     decides) resets `finished` and `front`, moves `epoch` on past whatever
 *     granules the dead launch left behind, and re-arms the word as
 *     {serial, 1} (it keeps ticket 0); the others draw again. A launch
 *     therefore never inherits the state of a predecessor that was aborted or
 *     whose bounded waits gave up (`detail::draw_ticket`).
 */
struct trsv_state {
    unsigned long long tickets;
    unsigned int finished;
    unsigned int epoch;
    // `kernel::trsv_wide` only: number of leading 128-row blocks whose
    // solution the next block's poller has read back from memory (monotone
    // within a solve, 0 between solves). Workgroups far behind the chain's front wait on this
    // one word instead of re-reading the granules they need.
    unsigned int front;
    unsigned int reserved;
};

constexpr unsigned int trsv_first_epoch = 2;
constexpr std::uint32_t trsv_clean_key = 0xC1EA0000u;
constexpr std::uint32_t trsv_arming_bit = 0x80000000u;
constexpr unsigned long long trsv_clean_tickets =
    static_cast<unsigned long long>(trsv_clean_key) << 32;

/** Kernel argument: where the state and the granules live. */
struct trsv_session {
    trsv_state *state;
    granule_word *granules;
    index_type capacity_words;  // granule words in the workspace
    std::uint32_t serial;       // this launch's key (never 0 or trsv_clean_key)
};

/** Host side: a key for the next launch (process-wide, so that launches of
 *  different streams on a shared workspace differ as well). A hipGraph replays
 *  its launch with the key it was captured with: fine, every completed solve
 *  leaves the clean key behind. */
inline std::uint32_t trsv_next_serial()
{
    static std::atomic<std::uint32_t> counter{1};
    std::uint32_t serial = counter.fetch_add(1, std::memory_order_relaxed);
    while (serial == 0 || serial == trsv_clean_key) {
        serial = counter.fetch_add(1, std::memory_order_relaxed);
    }
    return serial;
}


/** Upper solves run as lower solves on the mirrored index l -> n - 1 - l. A
 *  tile row of logical columns [c, c + w) then starts at memory column
 *  n - c - w: on a cache-line boundary only when n is a multiple of 128 bytes
 *  of storage, otherwise every tile row lies across one more line (the CLI's
 *  default upper sweep, n = 100, 200, ...: 12100 took 208 us where 12000 took
 *  180). So the vector paths of both kernels take the mirror around a virtual
 *  size n + pad that IS such a multiple: the first `pad` logical unknowns do
 *  not exist (identity rows, right-hand side 0; their x = 0 is handed on like
 *  any other, nothing is loaded or stored for them), every tile is aligned,
 *  and the row blocks are those of the virtual size. */
template <typename StType>
constexpr index_type trsv_virtual_pad(bool upper, index_type n)
{
    constexpr index_type line = 128 / static_cast<index_type>(sizeof(StType));
    return upper ? (line - n % line) % line : index_type{0};
}


/**
 * Whether the sweeps may read the rows of `mtx` with 16-byte loads. The
 * hardware needs no alignment for them (the queues run in unaligned mode; the
 * compiler emits the same `global_load_dwordx4` whatever alignment the pointer
 * type promises), so any base and any row stride qualify. A load that is not
 * 16-byte aligned may lie across two cache lines and is then served as two
 * requests - with an odd row stride that is the case for some rows and not
 * for others -, which costs a few per cent; the element-wise kernels (Vec = 1)
 * that served such operands until round 3 issued four to eight times the
 * load instructions and were 48 % slower at n = 24000 (fp32 storage,
 * lda = n + 1: profiles/r04_trsv_odd_strides.txt). `any_alignment` = false
 * restores that rule (option "trsv_vector_rows" = 0).
 */
template <typename MtxView>
bool trsv_vector_rows(const MtxView &mtx, bool any_alignment)
{
    using mtx_traits = view_traits<MtxView>;
    using st_type = std::remove_cv_t<typename mtx_traits::storage_type>;
    constexpr int vec16 = 16 / sizeof(st_type);
    if (any_alignment) {
        return is_aligned(mtx_traits::data(mtx), sizeof(st_type));
    }
    return is_aligned(mtx_traits::data(mtx), 16) &&
           mtx_traits::stride(mtx) % vec16 == 0;
}


namespace kernel {


namespace detail {


// Every spin is bounded: a wave that polled `trsv_spin_limit` times without
// success (seconds; a healthy solve needs microseconds to milliseconds) writes
// `trsv_error_word` into helper[0] and gives up; every other poller sees that
// word within 1024 polls and gives up as well, so a protocol bug ends in a
// wrong result that the caller can detect (helper[0] != ~0u) instead of a hung
// GPU.
constexpr unsigned int trsv_spin_limit = 1u << 21;
// Last-resort bound of a wait that polls LDS only (the relay variant's workers
// waiting for the loader wave). The loader is the one party that declares a
// solve failed - its waits on memory are bounded by `trsv_spin_limit`, and it
// raises `relay_abort` when it gives up, which is what the workers leave their
// wait on. Their own bound only guards against a loader that never says
// anything, and must not expire before the loader's can: the loader's worst case
// is `trsv_spin_limit` polls of <= ~2 us each (a round trip under load plus its
// s_sleep) = ~4 s over a whole solve; a worker looks at LDS once per ~150 cycles
// (s_sleep(1) + the read), 2^30 looks = ~67 s, counted afresh for every panel.
// A solve stretched by other streams' kernels therefore completes, as it does
// in the prefetching workgroups (`prefetch_ahead`).
constexpr unsigned int trsv_lds_spin_limit = 1u << 30;
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

/** What a workgroup learns when it joins a solve. */
struct trsv_ticket {
    std::uint32_t ticket;  // ~0u: the wait for a usable ticket gave up
    std::uint32_t epoch;   // tag of this solve's granules
};

/**
 * Draws this workgroup's ticket (one thread calls it) and reads the solve's
 * epoch. The common case is ONE atomic add on `trsv_state::tickets`: a word
 * that carries the clean key or this launch's serial is in order. Anything
 * else was left by a launch that did not finish, or is garbage: the callers
 * then agree through a compare-and-swap on one of them that repairs the state
 * (see `trsv_state`) while the others wait for the re-armed word. Waits are
 * bounded like every other wait of the solve (`spin_expired`).
 */
__device__ __forceinline__ trsv_ticket draw_ticket(
    trsv_state *state, std::uint32_t serial, std::uint32_t *helper,
    granule_word *granules, index_type capacity_words)
{
    using word = unsigned long long;
    const auto key_of = [](word w) { return static_cast<std::uint32_t>(w >> 32); };
    const auto low_of = [](word w) { return static_cast<std::uint32_t>(w); };
    const word armed = static_cast<word>(serial) << 32;
    trsv_ticket out{~std::uint32_t{0}, 0};
    const word old = __hip_atomic_fetch_add(&state->tickets, word{1},
                                            __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    if (old == trsv_clean_tickets) {
        // first of a clean state: stamp the launch's serial into the key (it
        // commutes with the other workgroups' draws; nobody waits for it)
        __hip_atomic_fetch_add(&state->tickets, armed - trsv_clean_tickets,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out.ticket = 0;
    } else if ((key_of(old) == trsv_clean_key || key_of(old) == serial) &&
               low_of(old) < trsv_arming_bit) {
        out.ticket = low_of(old);
    } else {
        // stale or overwritten header
        unsigned int spins = 0;
        word seen = old;
        while (true) {
            if (key_of(seen) == serial) {
                if (low_of(seen) < trsv_arming_bit) {
                    // re-armed by somebody else: draw again
                    const word again = __hip_atomic_fetch_add(
                        &state->tickets, word{1}, __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);
                    if (key_of(again) == serial &&
                        low_of(again) < trsv_arming_bit) {
                        out.ticket = low_of(again);
                        break;
                    }
                    seen = again;
                    continue;
                }
                // being repaired: wait
                if (spin_expired(spins, helper)) {
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                seen = __hip_atomic_load(&state->tickets, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            word expected = seen;
            if (__hip_atomic_compare_exchange_strong(
                    &state->tickets, &expected, armed | trsv_arming_bit,
                    __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT)) {
                // sole repairer: a fresh epoch (granules of the dead launch
                // must not be taken for this one's), counters at zero, then
                // the word re-armed with ticket 0 taken
                std::uint32_t epoch =
                    __hip_atomic_load(&state->epoch, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT) +
                    1;
                epoch = epoch < trsv_first_epoch ? trsv_first_epoch : epoch;
                // The epoch word of a broken header proves nothing (bytes
                // written over it may have moved it BACK, all-ones wraps to
                // the first epoch): hand-off words of an earlier solve could
                // carry the new tag. So the repairer clears every word, as the
                // end of a solve does when the epoch wraps. One thread, up to
                // a few MB of fire-and-forget stores: a millisecond on a path
                // that only a dead launch or a scribbler leads to.
                for (index_type i = 0; i < capacity_words; ++i) {
                    __hip_atomic_store(granules + i, granule_word{0},
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(&state->finished, 0u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&state->front, 0u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&state->epoch, epoch, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                drain_vmem();  // performed before the word says "armed"
                __hip_atomic_store(&state->tickets, armed | word{1},
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.ticket = 0;
                break;
            }
            seen = expected;  // what the compare-and-swap found instead
        }
    }
    out.epoch = __hip_atomic_load(&state->epoch, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
    return out;
}

/** The last workgroup of a solve (one thread calls it): the state as the
 *  next solve expects it. */
__device__ __forceinline__ void rearm_state(trsv_state *state,
                                            std::uint32_t next_epoch)
{
    __hip_atomic_store(&state->finished, 0u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&state->front, 0u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&state->epoch, next_epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&state->tickets, trsv_clean_tickets, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
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

/** One 16 x 16 x 4 matrix-core step D = A B + C in T (v_mfma_f64_16x16x4_f64 /
 *  v_mfma_f32_16x16x4_f32). Lane l supplies A[l & 15][l >> 4] and
 *  B[l >> 4][l & 15] and holds D[row(l, reg)][l & 15], reg = 0..3. */
template <typename T>
struct mma_16x16x4;

template <>
struct mma_16x16x4<double> {
    using acc_type = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    __device__ __forceinline__ static acc_type step(double a, double b,
                                                    acc_type c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __device__ __forceinline__ static int row(int lane, int reg)
    {
        return (lane >> 4) + 4 * reg;
    }
};

template <>
struct mma_16x16x4<float> {
    using acc_type = __attribute__((__vector_size__(4 * sizeof(float)))) float;
    __device__ __forceinline__ static acc_type step(float a, float b,
                                                    acc_type c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ __forceinline__ static int row(int lane, int reg)
    {
        return (lane >> 4) * 4 + reg;
    }
};

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

    /** `tag` marks the granule as published in THIS solve: the solve's epoch
     *  (>= 2; the workspace is never zeroed between solves). */
    __device__ __forceinline__ static void publish(granule_word *granules,
                                                   index_type element, St value,
                                                   std::uint32_t tag = 1)
    {
        const granule_word high = granule_word{tag} << 32;
        if (words == 1) {
            unsigned int bits = 0;
            __builtin_memcpy(&bits, &value, sizeof(St));
            __hip_atomic_store(granules + element, high | bits,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            granule_word bits = 0;
            __builtin_memcpy(&bits, &value, sizeof(St));
            __hip_atomic_store(granules + 2 * element,
                               high | (bits & 0xffffffffull),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(granules + 2 * element + 1,
                               high | (bits >> 32), __ATOMIC_RELAXED,
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
        const granule_word (&raw)[words], std::uint32_t tag = 1)
    {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < words; ++w) {
            ok = ok && (raw[w] >> 32) == tag;
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



/** K-loop of one 16 x 16 output tile on the matrix cores:
 *  sum_{k in [k_begin, k_end)} A[.][k] B[k][.], k_end - k_begin a multiple of
 *  16. `a_of(k)` / `b_of(k)` return this lane's operand for column / row k of
 *  the step (lane l supplies k = k0 + (l >> 4), see mma_16x16x4); the eight
 *  operand fetches of a pass are issued before its four dependent MFMAs. */
template <typename T, typename AFetch, typename BFetch>
__device__ __forceinline__ typename mma_16x16x4<T>::acc_type mma_tile_product(
    int k_begin, int k_end, int frag_k, AFetch a_of, BFetch b_of)
{
    using mma = mma_16x16x4<T>;
    typename mma::acc_type acc{};
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
        T a[4];
        T b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = a_of(k0 + q * 4 + frag_k);
            b[q] = b_of(k0 + q * 4 + frag_k);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = mma::step(a[q], b[q], acc);
        }
    }
    return acc;
}


/**
 * Forward substitution with a 64 x 64 lower-triangular tile by ONE wave: lane i
 * owns row i and its running right-hand side `r`, the solved x_j goes from lane
 * j to all lanes through SGPRs (`v_readlane`), one fused multiply-add per step
 * and lane; a row's coefficients come from LDS sixteen columns ahead. ~35
 * cycles per unknown - against 16 serial steps, two levels of matrix-core
 * products and four barriers for an inverse, which pays only where a block
 * waits for its turn with the inverse in hand.
 * T[i][j] = tile[i * row_stride + j * col_stride]; entries on and above the
 * diagonal are not read. `inv_diag` = 1 / T[i][i] (1 for a unit diagonal).
 * Rows from `rows` on do not exist. Returns x_i in lane i:
 * x_i = (r_i - sum_{j < i} T[i][j] x_j) * inv_diag_i, the sum in ascending j.
 */
template <bool UnitDiag, typename T>
__device__ __forceinline__ T substitute_tile64(const T *tile, int row_stride,
                                               int col_stride, T r, T inv_diag,
                                               int lane, int rows)
{
    constexpr int chunk = 16;
#pragma unroll
    for (int j0 = 0; j0 < trsv_tile; j0 += chunk) {
        if (j0 < rows) {  // (wave-uniform)
            // this lane's coefficients of columns j0 .. j0 + 15: all LDS reads
            // of the chunk are issued before its first dependent step
            T a[chunk];
#pragma unroll
            for (int q = 0; q < chunk; ++q) {
                const T value =
                    tile[lane * row_stride + (j0 + q) * col_stride];
                a[q] = lane > j0 + q ? value : T{};
            }
#pragma unroll
            for (int q = 0; q < chunk; ++q) {
                const T mine = UnitDiag ? r : r * inv_diag;
                const T xj = read_lane(mine, j0 + q);
                r -= a[q] * xj;
            }
        }
    }
    return UnitDiag ? r : r * inv_diag;
}


/**
 * In-place inverse of a 64 x 64 lower-triangular tile in LDS by a team of (at
 * least) four waves; every thread of the team must call it (it contains team
 * barriers = __syncthreads(): all teams of the workgroup call it together).
 *
 * In: T row-major, `tile[r * ld + c]`, zeros above the diagonal, ld = 65.
 * Out: inv(T) column-major, `tile[c * ld + r]`, zeros above the diagonal.
 * For T = [T00 0; C T11]: inv(T) = [X00 0; -X11 C X00, X11]. The serial part
 * is only 16 rows long: the four 16-lane groups of team wave 0 invert the four
 * diagonal 16 x 16 blocks in one instruction stream (lanes own columns in
 * registers, X[i][j] = (delta_ij - sum_{k<i} T[i][k] X[k][j]) / T[i][i]). The
 * two merge levels (16 -> 32 -> 64) are small dense products on the matrix
 * cores; the intermediate C X00 passes through `scratch_a` / `scratch_b` (512
 * words each) to get from the MFMA result layout into the operand layout.
 * Every finished block of the inverse overwrites zeros or its own source, and
 * the off-diagonal source blocks are cleared once they have been used.
 * `team_wave` = index of the calling wave in its team (waves >= 4 only take
 * part in the barriers and the final clearing), `team_thread` / `team_size` =
 * index of the thread in the team and the team's thread count.
 */
template <bool UnitDiag, typename ar_type>
__device__ __forceinline__ void invert_tile64(ar_type *tile, ar_type *scratch_a,
                                              ar_type *scratch_b, int team_wave,
                                              int lane, int team_thread,
                                              int team_size)
{
    constexpr int ld = trsv_tile + 1;
    const int wave = team_wave;
    ar_type *const partial_a = scratch_a;
    ar_type *const partial_b = scratch_b;
    {
        using mma = mma_16x16x4<ar_type>;
        constexpr int Q = 16;  // base block / MFMA tile edge
        const int frag_ij = lane & 15;
        const int frag_k = lane >> 4;
        if (wave == 0) {
            const int off = frag_k * Q;  // this 16-lane group's block
            ar_type column[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                // a dependent fp64 FMA chain costs ~40 cycles per link on
                // gfx950: four interleaved partial sums per row
                ar_type sums[4] = {};
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    sums[k & 3] += tile[(off + i) * ld + off + k] * column[k];
                }
                ar_type value = ((frag_ij == i) ? ar_type{1} : ar_type{0}) -
                                ((sums[0] + sums[1]) + (sums[2] + sums[3]));
                if (!UnitDiag) {
                    value = value / tile[(off + i) * ld + off + i];
                }
                column[i] = value;
            }
            // all reads of the four diagonal blocks are done (same wave,
            // program order): overwrite them with their inverses
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                tile[(off + frag_ij) * ld + off + i] = column[i];
            }
        }
        __syncthreads();
        // level 1, waves 0 and 1: the 32 x 32 diagonal block at o = 32 * wave.
        // P = C X00 (16 x 16 x 16), through this wave's 256 words of
        // partial_a, then X10 = -X11 P.
        if (wave < 2) {
            const int o = 2 * Q * wave;
            ar_type *scratch = partial_a + wave * Q * Q;
            typename mma::acc_type p{};
#pragma unroll
            for (int k0 = 0; k0 < Q; k0 += 4) {
                const int k = k0 + frag_k;
                p = mma::step(tile[(o + Q + frag_ij) * ld + o + k],  // C[i][k]
                              tile[(o + frag_ij) * ld + o + k],      // X00[k][j]
                              p);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                scratch[mma::row(lane, reg) * Q + frag_ij] = p[reg];
            }
            typename mma::acc_type x10{};
#pragma unroll
            for (int k0 = 0; k0 < Q; k0 += 4) {
                const int k = k0 + frag_k;
                x10 = mma::step(
                    tile[(o + Q + k) * ld + o + Q + frag_ij],  // X11[i][k]
                    scratch[k * Q + frag_ij],                  // P[k][j]
                    x10);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                // X10[r][c] of the inverse, column-major
                tile[(o + frag_ij) * ld + o + Q + mma::row(lane, reg)] =
                    -x10[reg];
            }
            // C (rows o+16.., columns o..o+15, row-major) is used up; in the
            // inverse's layout that region is above the diagonal: zero
#pragma unroll
            for (int e = lane; e < Q * Q; e += wave_size) {
                tile[(o + Q + e / Q) * ld + o + e % Q] = ar_type{};
            }
        }
        __syncthreads();
        // level 2: X10 (32 x 32) = -X11 (C X00) with the two 32 x 32 inverses.
        // Waves 0..3 own one 16 x 16 tile each; P = C X00 is handed over
        // through partial_a (waves 0, 1) and partial_b (waves 2, 3).
        constexpr int H = 2 * Q;
        const int ti = (wave >> 1) & 1;
        const int tj = wave & 1;
        ar_type *const scratch2 =
            (wave < 2 ? partial_a : partial_b) + (wave & 1) * Q * Q;
        if (wave < 4) {
            typename mma::acc_type p{};
            // X00 is lower triangular: rows above column tile tj are zero
            for (int k0 = tj * Q; k0 < H; k0 += 4) {
                const int k = k0 + frag_k;
                p = mma::step(tile[(H + ti * Q + frag_ij) * ld + k],  // C[i][k]
                              tile[(tj * Q + frag_ij) * ld + k],      // X00[k][j]
                              p);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                scratch2[mma::row(lane, reg) * Q + frag_ij] = p[reg];
            }
        }
        __syncthreads();
        if (wave < 4) {
            typename mma::acc_type x10{};
            // X11 is lower triangular: columns beyond row tile ti are zero
            for (int k0 = 0; k0 < (ti + 1) * Q; k0 += 4) {
                const int k = k0 + frag_k;
                // P tile (k / 16, tj) was written by wave 2 * (k / 16) + tj
                const ar_type *src =
                    ((k / Q) == 0 ? partial_a : partial_b) + tj * Q * Q;
                x10 = mma::step(
                    tile[(H + k) * ld + H + ti * Q + frag_ij],  // X11[i][k]
                    src[(k % Q) * Q + frag_ij],                 // P[k][j]
                    x10);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                tile[(tj * Q + frag_ij) * ld + H + ti * Q +
                     mma::row(lane, reg)] = -x10[reg];
            }
        }
        __syncthreads();
        // C (rows >= 32, columns < 32, row-major) is used up: the upper-right
        // block of the inverse is zero
        for (int e = team_thread; e < H * H; e += team_size) {
            tile[(H + e / H) * ld + e % H] = ar_type{};
        }
    }
}


}  // namespace detail


/**
 * Blocked triangular solve. blockDim = 64 * Waves, grid = ceil(n / 64).
 *
 * Logical (always lower-triangular) indices i' map to memory indices by
 * `phys`: identity for Lower, n-1-i' for Upper. Logical block b covers
 * i' in [64 b, 64 b + 63]; entries with i' >= n do not exist (masked).
 * Granules are indexed by the logical element index.
 *
 * Vec = elements per 16-byte load in the far sweep (any base and row stride,
 * see `trsv_vector_rows`); Vec = 1 reads element by element.
 *
 * Meaning per block as in `cuda/trsv_kernels.cuh:527-692` (lower) /
 * `:725-893` (upper): x_b = inv(T_bb) * (x_b - sum_{c<b} A_bc x_c), all in
 * ArType, x stored as StType and re-read by later blocks as stored.
 */
template <int Waves, bool Upper, bool UnitDiag, int Vec, bool Relay,
          typename ArType, bool Persistent = false, typename MtxView,
          typename XView>
__global__ __launch_bounds__((Waves + (Relay ? 1 : 0)) * wave_size) void trsv(
    MtxView mtx, XView x, std::uint32_t *helper, trsv_session session)
{
    using ar_type = decltype(mtx(0, 0) * x(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value,
                  "accessor arithmetic must yield ArType");
#ifdef ACCBLAS_TRSV_ENTRY_NOPS
    // (tools/tune_trsv_align.hip only: shifts the kernel's code by 4 bytes per
    // s_nop to see what instruction alignment does to the solve)
#define ACCBLAS_TRSV_STR2(_x) #_x
#define ACCBLAS_TRSV_STR(_x) ACCBLAS_TRSV_STR2(_x)
    asm volatile(".rept " ACCBLAS_TRSV_STR(ACCBLAS_TRSV_ENTRY_NOPS) "\n s_nop 0\n .endr");
#endif
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
    __shared__ ar_type own_lds[B];  // this block's right-hand side
    __shared__ std::int32_t shared_block;
    constexpr int XK = 16;  // panels of x in the LDS ring of the far sweep
    __shared__ ar_type x_ring[XK][B];
    __shared__ std::int32_t x_ring_tag[XK];       // panel index + 1
    __shared__ std::int32_t sweep_progress[Waves];  // panels consumed
    // relay variant: set when the loader's bounded wait gave up or a worker
    // waited "for ever" on the ring: the workers' waits look at LDS only
    __shared__ std::int32_t relay_abort;

    const int lane_of_thread = lane_id();
    const int wave_of_thread = wave_id();
    // `n` = the size the blocks are laid out for (vector path of an upper
    // solve: `pad` virtual unknowns in front, see `trsv_virtual_pad`); logical
    // index l exists if pad <= l < n
    const index_type pad =
        Vec > 1 ? trsv_virtual_pad<st_type>(Upper, mtx.length(0))
                : index_type{0};
    const index_type n = mtx.length(0) + pad;
    const auto exists = [pad, n](index_type logical) {
        return logical >= pad && logical < n;
    };
    const auto inside = [pad, n](index_type logical) {
        return logical < pad ? pad : (logical < n ? logical : n - 1);
    };

    granule_word *const granules = session.granules;
    __shared__ std::uint32_t shared_epoch;
    __shared__ bool is_last;
    // Row blocks of the solve, and the events the LAST of which re-arms the
    // workspace: every row block done and every workgroup's final draw.
    const unsigned int num_blocks =
        static_cast<unsigned int>(ceildiv(n, index_type{B}));
    // (a launch with a workgroup for every row block - small systems - has
    // nobody draw a second ticket: the kernel would only end two atomic round
    // trips later)
    // (Persistent = false: the grid has a workgroup per row block; its code
    // has no loop around the row block, whose invariants the compiler would
    // compute up front and keep in registers, see below)
    constexpr bool persistent = Persistent;
    const unsigned int last_event =
        persistent ? num_blocks + gridDim.x - 1 : num_blocks - 1;
    constexpr std::int32_t no_block_left = 0x3ffffffe;
    constexpr std::int32_t no_ticket = 0x3fffffff;
    // The last thing a workgroup does with the workspace (every thread calls
    // it): counts the event, and the last one of the solve resets the
    // counters for its successor.
    const auto count_event = [&](std::uint32_t epoch_now) {
        if (threadIdx.x == 0) {
            const unsigned int done = __hip_atomic_fetch_add(
                &session.state->finished, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            is_last = done == last_event;
        }
        __syncthreads();
        if (is_last) {
            std::uint32_t next_epoch = epoch_now + 1;
            if (next_epoch == 0) {
                // 2^32 solves on this workspace: stale tags could repeat
                for (index_type i = threadIdx.x; i < session.capacity_words;
                     i += blockDim.x) {
                    __hip_atomic_store(granules + i, granule_word{0},
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                next_epoch = trsv_first_epoch;
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                detail::rearm_state(session.state, next_epoch);
            }
        }
    };
    // A workgroup solves row blocks until the tickets run out: the launch has
    // at most one workgroup per CU (`launch_trsv`), and a CU whose block is
    // done draws the next ticket at once. With one workgroup per ROW BLOCK the
    // blocks beyond the CU count wait for the hardware dispatcher, which hands
    // workgroup k to XCD k mod 8 whatever is free elsewhere: the row blocks
    // with the MOST far columns then start when the last XCD's first blocks
    // are through (n = 22100, 8-byte storage: ticket 336 of 346 drawn 207-310
    // us into a 480-590 us solve, with 10 MB to stream before the chain
    // arrives, while the CUs of seven XCDs idle; whether they made it decided
    // between two run-time modes 20 % apart, profiles/r04_trsv_plain_vs_acc.txt).
    for (;;) {
    // (lane and wave index are made opaque once per row block: everything
    // the body computes from them - a few hundred LDS and matrix addresses -
    // is invariant in this loop, and the compiler would otherwise compute all
    // of it in front of the loop and keep it in registers: 168 registers and
    // 200-590 bytes of scratch instead of 78-129 registers)
    int lane_now = lane_of_thread;
    int wave_now = wave_of_thread;
    if constexpr (persistent) {
        asm volatile("" : "+v"(lane_now), "+v"(wave_now));
    }
    const int lane = lane_now;
    const int wave = wave_now;
#ifdef ACCBLAS_TRSV_DIAG
    unsigned long long diag_stamp[16] = {};
#endif
    // waves 0 .. Waves-1 work on the block's rows; wave `Waves` is the loader:
    // it feeds the solutions of earlier blocks into LDS for the far sweep
    // (see there) and otherwise only takes part in the barriers
    const bool worker = !Relay || wave < Waves;
    if (threadIdx.x < XK) {
        x_ring_tag[threadIdx.x] = 0;
    }
    if (threadIdx.x < Waves) {
        sweep_progress[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        relay_abort = 0;
    }
    if (threadIdx.x == 0) {
        // the previous solve on this workspace finished before this kernel
        // started (stream order), so the epoch is stable for the whole solve
        const detail::trsv_ticket drawn =
            detail::draw_ticket(session.state, session.serial, helper,
                                session.granules, session.capacity_words);
        const unsigned int ticket = drawn.ticket;
        shared_epoch = drawn.epoch;
        // (a ticket that could not be had - ~0u -: the workgroup leaves
        // without a word, helper[0] says why; a ticket beyond the last row
        // block: nothing left to do)
        shared_block = ticket < num_blocks
                           ? static_cast<std::int32_t>(ticket)
                           : (ticket == ~0u ? no_ticket : no_block_left);
        if (ticket == 0) {
            // `trsv_init` (cuda/trsv_kernels.cuh:38-42) folded into the first
            // block: helper[0] stays ~0 unless a bounded spin expires (nobody
            // looks at it before 1024 unsuccessful polls), helper[1] mirrors
            // the reference's block counter after the solve
            __hip_atomic_store(helper, ~std::uint32_t{0}, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            helper[1] = num_blocks;
        }
    }
    __syncthreads();
    const index_type block = shared_block;
    const std::uint32_t epoch = shared_epoch;
    if (block == no_block_left) {
        count_event(epoch);
        return;
    }
    if (block == no_ticket) {
        return;
    }
    const index_type base = block * B;
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
    const bool own_valid = exists(own_row);
    st_type *own_addr = x_data + phys(inside(own_row)) * x_inc;
    // (kept in LDS, not in a register of every wave, until the far sweep is
    // over: the far sweep needs the registers)
    // (persistent: the load goes out ahead of the tile's, the value is parked
    // in LDS behind them - the loop around the row block needs the registers;
    // one block per workgroup: every wave keeps it in a register, 1 us faster
    // per solve at n = 200 ... 2000)
    st_type own_raw{};
    if (!persistent || wave == 0) {
        own_raw = *own_addr;  // (clamped address: unconditional)
    }
    const ar_type own_rhs =
        own_valid ? static_cast<ar_type>(own_raw) : ar_type{};

    // ---- diagonal tile and last panel -> LDS (a wave reads one row) -------
    // Every load of the wave's rows is issued before the first use, so the
    // staging costs one memory latency, not one per row: the loads are
    // UNCONDITIONAL (row and column clamped into the matrix; what lies above
    // the diagonal or beyond n is read and dropped) and stay in the storage
    // type until all are back - with the loads inside the conditions the
    // compiler had put a full wait behind each of the 16 (5 us per block).
    if (worker) {
        constexpr int RS = B / Waves;  // rows a wave stages
        st_type diag_raw[RS];
        st_type panel_raw[RS];
        const index_type gc = base + lane;
        const index_type col = phys(inside(gc));
        const index_type panel_gc = block > 0 ? base - B + lane : index_type{0};
        const index_type panel_col = phys(inside(panel_gc));
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const index_type gr = base + wave + i * Waves;
            const st_type *row = a_data + phys(inside(gr)) * lda;
            diag_raw[i] = row[col];
            panel_raw[i] = row[panel_col];
        }
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int r = wave + i * Waves;
            const index_type gr = base + r;
            ar_type diag_value{};
            if (lane == r) {
                diag_value = (UnitDiag || !exists(gr))
                                 ? ar_type{1}
                                 : static_cast<ar_type>(diag_raw[i]);
            } else if (lane < r && exists(gr) && exists(gc)) {
                diag_value = static_cast<ar_type>(diag_raw[i]);
            }
            tile[r * ld + lane] = diag_value;
            if (block > 0) {
                panel[lane * ld + r] =
                    exists(gr) && exists(panel_gc)
                        ? static_cast<ar_type>(panel_raw[i])
                        : ar_type{};
            }
        }
    }
    if (persistent && wave == 0) {
        own_lds[lane] = own_rhs;
    }
    __syncthreads();

    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 10);  // tile and panel staged
    }

    // ---- invert the tile (recursively; see detail::invert_tile64) -----------
    static_assert(Waves >= 8, "the partial-sum buffers double as the 4 x 256 "
                              "words of scratch of the inversion");
    detail::invert_tile64<UnitDiag>(tile, partial_a, partial_b, wave, lane,
                                    static_cast<int>(threadIdx.x),
                                    static_cast<int>(blockDim.x));
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 1);  // inverse ready
    }
    __syncthreads();

    // ---- fold the inverse into the last panel: M = inv(T_bb) * A_{b,b-1} ---
    // x_b = inv(T) (rhs_far - A_{b,b-1} x_{b-1}) = inv(T) rhs_far - M x_{b-1}:
    // with M prepared here, off the critical path, only ONE mat-vec is left
    // between the arrival of x_{b-1} and the publication of x_b. This is the
    // one dense 64 x 64 x 64 product of the kernel, so it runs on the matrix
    // cores: 16 output tiles of 16 x 16 spread over the waves, 16 x 16 x 4
    // MFMA steps in the arithmetic type (exact fp64 / fp32 FMAs), operands
    // fetched from LDS one element per lane. inv(T) is lower triangular, so a
    // row tile stops at its own diagonal tile.
    constexpr int CW = B / Waves;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (block > 0) {
        using mma = detail::mma_16x16x4<ar_type>;
        constexpr int tiles_per_wave = 16 / Waves;
        static_assert(tiles_per_wave >= 1 && 16 % Waves == 0,
                      "16 output tiles must split evenly over the waves");
        const int frag_ij = lane & 15;  // row of A / column of B
        const int frag_k = lane >> 4;
        // tile (ti, tj) of M costs ti + 1 passes over 16 columns of inv(T):
        // with 8 waves a wave takes row tiles {0, 3} or {1, 2} of one column
        // tile, five passes each
        const auto tile_of = [&](int t, int &ti, int &tj) {
            if (Waves == 8) {
                tj = wave_u % 4;
                ti = (t == 0) ? wave_u / 4 : 3 - wave_u / 4;
            } else {
                ti = (wave_u * tiles_per_wave + t) / 4;
                tj = (wave_u * tiles_per_wave + t) % 4;
            }
        };
        typename mma::acc_type m_tile[tiles_per_wave];
#pragma unroll
        for (int t = 0; t < tiles_per_wave; ++t) {
            int ti, tj;
            tile_of(t, ti, tj);
            typename mma::acc_type acc{};
            for (int pass = 0; worker && pass <= ti; ++pass) {
                // all eight operand loads of a pass are issued before the
                // first of its four dependent MFMAs
                ar_type a[4];
                ar_type b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = pass * 16 + q * 4 + frag_k;
                    a[q] = tile[k * ld + ti * 16 + frag_ij];
                    b[q] = panel[(tj * 16 + frag_ij) * ld + k];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc = mma::step(a[q], b[q], acc);
                }
            }
            m_tile[t] = acc;
        }
        __syncthreads();  // every wave is done reading A_{b,b-1}
#pragma unroll
        for (int t = 0; t < tiles_per_wave; ++t) {
            int ti, tj;
            tile_of(t, ti, tj);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (worker) {
                    panel[(tj * 16 + frag_ij) * ld + ti * 16 +
                          mma::row(lane, reg)] = m_tile[t][reg];
                }
            }
        }
        // readers of M wait behind the barrier that ends the far sweep
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
    // x reaches the workers through LDS: only the loader wave reads the
    // granules of a panel (lane = element, ONE load instruction per panel
    // and workgroup, several panels in flight), decodes them into a ring of
    // XK panels and tags the slot; the workers wait on the LDS tag. Every
    // worker polling every granule itself cost as many bytes as the matrix
    // (8-byte uncached loads, 16 KB per panel and workgroup): at n = 65536
    // that second stream, not the chain, set the pace (2 x 3.8 TB/s through
    // the fabric). The loader has no matrix loads of its own, so it runs
    // ahead of the workers by up to XK panels and they stay independent of
    // each other.
    // Small systems keep the direct variant (Relay = false, no loader wave,
    // every worker polls the granules it needs): the detour costs ~0.1 us per
    // hop when the chain is all there is (n = 8192: 172 vs 190 us).
    struct stage {
        st_type a[NI][Vec];
        granule_word xraw[Relay ? 1 : Vec][io::words];  // direct variant only
    };
    const st_type *row_ptr[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const index_type gr = base + wave * RW + k * Vec + row_in_inst;
        row_ptr[k] = a_data + phys(inside(gr)) * lda;
    }
    // (upper: first logical column from which whole vectors exist)
    const index_type first_real_vector = ceildiv(pad, index_type{Vec}) * Vec;
    const auto issue = [&](index_type cb, stage &s) {
        const index_type first_col = cb * B + col_group * Vec;  // logical
        // lowest memory column of the group (columns descend for Upper;
        // vectors of virtual columns only read the first vector of real ones
        // instead - their x is 0 -; the one vector that holds both, when n is
        // not a multiple of Vec, is cleaned up in `consume`)
        const index_type mem_col =
            Upper ? n - 1 -
                        ((first_col + Vec <= pad ? first_real_vector
                                                 : first_col) +
                         Vec - 1)
                  : first_col;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (Vec > 1) {
                const auto v = load_packed<(Vec > 1 ? Vec : 2), ACCBLAS_TRSV_NT>(
                    row_ptr[k] + mem_col);
#pragma unroll
                for (int e = 0; e < Vec; ++e) {
                    s.a[k][e] = v[e];
                }
            } else {
                s.a[k][0] = row_ptr[k][mem_col];
            }
        }
        if constexpr (!Relay) {
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                io::load(granules, first_col + e, s.xraw[e]);
            }
        }
    };
    ar_type acc[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        acc[k] = ar_type{};
    }
    unsigned int spins = 0;
    const auto consume = [&](index_type cb, stage &s) {
        const int slot = static_cast<int>(cb % XK);
        const std::int32_t wanted = static_cast<std::int32_t>(cb) + 1;
        ar_type xv[Vec];
        if constexpr (!Relay) {
            const index_type first_col = cb * B + col_group * Vec;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < Vec; ++e) {
                    ok = ok && io::ready(s.xraw[e], epoch);
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
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                xv[e] = static_cast<ar_type>(io::decode(s.xraw[e]));
            }
        } else {
            // The wait looks at LDS and nothing else. A worker's vector-memory
            // queue must hold its matrix loads ONLY: with `spin_expired`'s
            // load of helper[0] in this loop the compiler cannot count what
            // is outstanding and puts `s_waitcnt vmcnt(0)` behind every
            // panel's FMAs - the wave then waits for the panel it has just
            // requested, the ring is one panel deep in time whatever its
            // depth on paper, and a row block that starts late (more blocks
            // than CUs) streams 32 KB per memory latency, too slow to be
            // ready when the chain reaches it (profiles/r04_trsv_relay_ring.txt).
            // The loader (which has no matrix loads) keeps the bounded wait
            // on memory and raises `relay_abort` when it gives up.
            unsigned int lds_spins = 0;
            while (lds_flag_load(&x_ring_tag[slot]) != wanted) {
                if (lds_flag_load(&relay_abort) != 0 ||
                    ++lds_spins >= detail::trsv_lds_spin_limit) {
                    lds_flag_store(&relay_abort, 1);
                    break;
                }
                if (wave == 0) {
                    ACCBLAS_TRSV_FIRST_WAIT(cb);
                }
                // eight waves spinning on LDS at full rate would take issue
                // slots from the loader on the same SIMDs
                __builtin_amdgcn_s_sleep(1);
            }
            lds_acquire();
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                xv[e] = x_ring[slot][col_group * Vec + e];
            }
        }
        if (Upper && Vec > 1 && cb == 0 && pad % Vec != 0) {
            // the vector across the first real column: what was loaded for
            // its virtual columns lies beyond the row's end
#pragma unroll
            for (int e = 0; e < Vec; ++e) {
                if (col_group * Vec + (Vec - 1 - e) < pad &&
                    col_group * Vec + Vec > pad) {
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        s.a[k][e] = st_type{};
                    }
                }
            }
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
        // this wave is done with the slot: the release keeps the compiler from
        // sinking the x_ring reads (or the FMAs that consume them) below the
        // progress store, after which the loader may overwrite the slot; LDS
        // serves a wave's operations in order
        if (Relay) {
            lds_release();
            if (lane == 0) {
                lds_flag_store(&sweep_progress[wave], wanted);
            }
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
    if (!worker) {
        // ---- loader wave: granules of panels [0, far_end) -> LDS ring -------
        constexpr int L = 8;  // panels whose granule loads are in flight
        static_assert(XK >= 2 * L, "room for the loader to run ahead");
        const index_type relayed = far_end;
        for (index_type p0 = 0; p0 < relayed; p0 += L) {
            granule_word raw[L][io::words];
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if (p0 + i < relayed) {
                    io::load(granules, (p0 + i) * B + lane, raw[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const index_type p = p0 + i;
                if (p < relayed) {
                    const int slot = static_cast<int>(p % XK);
                    const std::int32_t wanted = static_cast<std::int32_t>(p) + 1;
                    // the slot still holds panel p - XK: wait until every
                    // worker has read it
                    bool gave_up = false;
                    if (p >= XK) {
                        const std::int32_t need = wanted - XK;
                        while (!__all(lane >= Waves ||
                                      lds_flag_load(&sweep_progress[lane < Waves ? lane : 0]) >=
                                          need)) {
                            if (lds_flag_load(&relay_abort) != 0 ||
                                detail::spin_expired(spins, helper)) {
                                gave_up = true;
                                break;
                            }
                        }
                        lds_acquire();
                    }
                    while (!gave_up && !__all(io::ready(raw[i], epoch))) {
                        if (detail::spin_expired(spins, helper)) {
                            gave_up = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        io::load(granules, p * B + lane, raw[i]);
                    }
                    if (gave_up) {
                        lds_flag_store(&relay_abort, 1);  // the workers stop waiting
                    }
                    x_ring[slot][lane] =
                        static_cast<ar_type>(io::decode(raw[i]));
                    // LDS operations of one wave complete in order: data,
                    // then the tag
                    lds_release();
                    if (lane == 0) {
                        lds_flag_store(&x_ring_tag[slot], wanted);
                    }
                }
            }
        }
    } else if (far_end > 0) {
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
        index_type cb0 = 0;
        if constexpr (Relay) {
            // Whole groups of `depth` panels that all have a successor in the
            // ring: no condition around the refill, so a slot's new loads land
            // in the slot's own registers (a conditional refill is a phi of
            // "loaded" and "kept": the compiler loads into other registers and
            // copies them over at the join, behind a full wait), and the refill
            // is kept BEHIND the panel's FMAs (`order_after`: issued ahead of
            // them it would need other registers again). The waits in front of
            // a panel's FMAs then read `vmcnt((depth - 1) * NI)`: the ring is
            // `depth` panels deep in time (tests/test_isa_guard.py).
            for (; cb0 + 2 * depth <= far_end; cb0 += depth) {
#pragma unroll
                for (int d = 0; d < depth; ++d) {
                    consume(cb0 + d, ring[d]);
                    index_type next = cb0 + d + depth;
                    order_after(next, acc);
                    issue(next, ring[d]);
                }
            }
        }
        for (; cb0 < far_end; cb0 += depth) {
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
    {
        // NI sums per lane over the LPR lanes of a row group: a transposing
        // butterfly on DPP moves (lane k of a group ends with the total of sum
        // k) instead of NI x log2(LPR) shuffles through the LDS crossbar,
        // which sat on every hop of the chain
        const ar_type total = transpose_reduce_sum<NI, LPR>(acc, lane);
        if (worker && col_group < NI) {
            const int r = wave * RW + col_group * Vec + row_in_inst;
            correction[r] = exists(base + r) ? total : ar_type{};
        }
    }
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 2);  // wave 0 finished its far sweep share
    }
    __syncthreads();
    if (Relay && threadIdx.x == 0 && lds_flag_load(&relay_abort) != 0) {
        // (whoever gave up: the caller must see that the solution is invalid)
        __hip_atomic_store(helper, detail::trsv_error_word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- tail: c = inv(T) rhs_far (before the wait), then x = c - M x_{b-1} -
    // Both are 64 x 64 mat-vecs from LDS with lane = row. The columns are
    // split over the waves (CW each); a wave broadcasts its column's vector
    // entry from the lane that holds it with v_readlane (no LDS round trip),
    // and the per-wave partial results are combined through LDS in a fixed
    // order.
    if (wave == 0) {
        ACCBLAS_TRSV_STAMP(block, 3);  // all waves finished the far sweep
    }
    ar_type rhs =
        own_valid ? (persistent ? own_lds[lane] : own_rhs) - correction[lane]
                  : ar_type{};
    if (worker) {
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
            while (!__all(io::ready(raw, epoch)) &&
                   !detail::spin_expired(spins, helper)) {
                io::load(granules, prev, raw);
            }
            ACCBLAS_TRSV_STAMP(block, 4);  // x_{b-1} observed
            ACCBLAS_TRSV_CYCLES(block, 11);
            x_prev_lds[lane] = static_cast<ar_type>(io::decode(raw));
        }
        __syncthreads();
        if (wave == 0) {
            ACCBLAS_TRSV_STAMP(block, 7);  // x_{b-1} handed to all waves
        }
        if (worker) {
            ar_type part[4] = {};
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                part[j & 3] += panel[(wave_u * CW + j) * ld + lane] *
                               x_prev_lds[wave_u * CW + j];
            }
            partial_a[wave_u * B + lane] =
                (part[0] + part[1]) + (part[2] + part[3]);
        }
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
        if (own_row < n) {
            // (virtual unknowns of an upper solve: handed on - x = 0 -, not
            // stored)
            io::publish(granules, own_row, solution, epoch);  // hand-off
            if (own_valid) {
                *own_addr = solution;  // the user's vector
            }
        }
        ACCBLAS_TRSV_STAMP(block, 5);  // published
        ACCBLAS_TRSV_CYCLES(block, 12);
    }
#ifdef ACCBLAS_TRSV_DIAG
    if (threadIdx.x == 0 && trsv_diag_buffer != nullptr) {
        for (int i = 0; i < 16; ++i) {
            trsv_diag_buffer[block * 16 + i] = diag_stamp[i];
        }
    }
#endif
    // ---- this block is done with the workspace (off the critical path) ----
    count_event(epoch);
    if constexpr (!persistent) {
        return;
    }
    __syncthreads();  // the next block re-initialises what lives in LDS
    }  // for (;;)
}


}  // namespace kernel


/** Helpers a split row block of the 128-row kernel can have (one slot of 128
 *  partial sums each in the workspace). */
#ifndef ACCBLAS_TRSV_PARTIAL_SLOTS
#define ACCBLAS_TRSV_PARTIAL_SLOTS 15
#endif
constexpr int trsv_partial_slots = ACCBLAS_TRSV_PARTIAL_SLOTS;

/** Bytes of hand-off workspace a solve of n unknowns needs behind the state
 *  header: one or two granule words per element of x, then (128-row kernel,
 *  split row blocks) the partial-sum area: `trsv_partial_slots` x 128
 *  two-word slots per 128-row block, 64-word aligned. */
template <typename St>
constexpr std::size_t trsv_workspace_bytes(index_type n)
{
    // (an upper solve of the 128-row kernel is laid out for up to 63 virtual
    // unknowns in front of the real ones, `trsv_wide_virtual_pad`)
    const index_type laid_out = n + 64;
    const index_type x_words =
        ceildiv(laid_out * trsv_granule_words<St>(), index_type{64}) * 64;
    const index_type partial_words =
        ceildiv(laid_out, index_type{128}) * trsv_partial_slots * 128 * 2;
    return static_cast<std::size_t>(x_words + partial_words) *
           sizeof(granule_word);
}


/**
 * Dynamic LDS requested on top of the kernel's static arrays so that exactly
 * ONE workgroup is resident per CU (160 KiB of LDS). A second resident block
 * puts its far sweep on the same SIMDs, LDS and memory pipes as the block
 * that is on the critical chain: measured with two per CU, n = 65536 took
 * 2.65 ms instead of 2.32 ms and fp64 storage at n = 24000 1.82 ms instead of
 * 0.91 ms (round-1 A/B of the one-block and two-block builds; the log was not
 * kept, profiles/r01_trsv_relay_ab.txt is from the one-block build).
 */
template <typename ArType, int Waves, bool Relay>
constexpr unsigned trsv_single_block_pad()
{
    constexpr std::size_t B = trsv_tile;
    // tile, panel, correction, x_prev, the two partial-sum buffers, a few
    // words of bookkeeping and, in the relay variant only (the compiler drops
    // it from the direct one), the x ring of the far sweep (16 panels); an
    // underestimate is harmless (more padding), an overestimate is not
    constexpr std::size_t static_bytes =
        sizeof(ArType) *
            (2 * B * (B + 1) + 3 * B + 2 * Waves * B + (Relay ? 16 * B : 0)) +
        16;
    constexpr std::size_t needed = 160 * 1024 / 2 + 2048;
    return static_bytes >= needed
               ? 0u
               : static_cast<unsigned>(needed - static_bytes);
}


/** When the far sweep's x goes through the loader wave and LDS. `relay_mode`
 *  of `launch_trsv` (option "trsv_relay" of accblas_set_option): 0 never, 1 by
 *  this rule, 2 always. 2-byte storage and fp32 arithmetic: always (round 1,
 *  profiles/r01_trsv_relay_ab.txt). fp64 arithmetic: from these sizes on. In
 *  round 1 the relay lost there at every size; its workers then waited for each
 *  panel they had just requested (see `consume`). With the ring in flight
 *  (round 4, profiles/r04_trsv_relay_ring.txt, us, relay | direct): 8-byte
 *  storage 512 18.2 | 17.6, 1024 27.2 | 29.2, 1500 35.4 | 44.2, 2048 44.5 |
 *  52.9, 8192 160 | 222, 16384 329 | 461; 4-byte storage 1024 26.3 | 24.7, 2048
 *  42.9 | 44.2, 3000 57.7 | 65.6, 4096 72.1 | 74.9 (from there the 128-row
 *  kernel takes over), 24000 431 | 475, 65536 1614 | 2208. Round 5, finer steps
 *  inside the reference's 24000-wide matrix (profiles/r05_trsv_relay_threshold.txt,
 *  relay | direct): 4-byte storage 1024 25.4 | 24.8, 1280 29.7 | 30.5, 1536 33.5 |
 *  35.4, 2000 42.4 | 45.8; 8-byte storage 896 24.8 | 24.5, 960 25.8 | 27.4, 1000
 *  26.9 | 27.8 - the thresholds moved from 2048 / 1024 to 1152 / 960 (the sweep's
 *  n = 2000 and n = 1000 were 4 % slower than the next size). */
#ifndef ACCBLAS_TRSV_RELAY_MIN_N
#define ACCBLAS_TRSV_RELAY_MIN_N 1152
#endif
#ifndef ACCBLAS_TRSV_RELAY_MIN_N_8BYTE
#define ACCBLAS_TRSV_RELAY_MIN_N_8BYTE 960
#endif
constexpr index_type trsv_relay_min_n = ACCBLAS_TRSV_RELAY_MIN_N;

/** Workgroups (= tickets) a solve of n unknowns launches. */
inline unsigned trsv_num_blocks(index_type n)
{
    return static_cast<unsigned>(ceildiv(n, index_type{trsv_tile}));
}


template <typename ArType, typename MtxView, typename XView>
hipError_t launch_trsv(bool upper, bool unit_diag, const MtxView &mtx,
                       const XView &x, std::uint32_t *helper,
                       const trsv_session &session, hipStream_t stream,
                       int relay_mode = 1, int num_cus = 0,
                       bool any_alignment = true)
{
    using mtx_traits = view_traits<MtxView>;
    using st_type =
        std::remove_cv_t<typename view_traits<XView>::storage_type>;
    static_assert(mtx_traits::is_pointer_view &&
                      view_traits<XView>::is_pointer_view,
                  "the solve streams the triangle with loads of the storage "
                  "type: reduced_row_major / row_major / plain views only");
    const index_type n = mtx.length(0);
    if (n <= 0) {
        return hipSuccess;
    }
    constexpr int waves = trsv_waves;
    constexpr int vec16 = 16 / sizeof(st_type);
    // x through the loader wave and LDS (one more wave per workgroup) where
    // it was measured to pay: the granule stream it removes weighs 8 bytes per
    // element and worker, and the workers' memory queues hold nothing but
    // their matrix panels (see `trsv_relay_min_n`)
    constexpr bool half_storage = sizeof(st_type) == 2;
    constexpr bool double_storage = sizeof(st_type) == 8;
    constexpr bool float_arithmetic = std::is_same<ArType, float>::value;
    const bool by_rule =
        half_storage || float_arithmetic ||
        n >= (double_storage ? index_type{ACCBLAS_TRSV_RELAY_MIN_N_8BYTE}
                             : trsv_relay_min_n);
    const bool relay = relay_mode == 2 || (relay_mode == 1 && by_rule);
    const dim3 block((waves + (relay ? 1 : 0)) * wave_size, 1, 1);
    const unsigned pad = relay ? trsv_single_block_pad<ArType, waves, true>()
                               : trsv_single_block_pad<ArType, waves, false>();
    // (upper solves of any n: the vector path mirrors around a virtual size)
    const bool vec_ok = trsv_vector_rows(mtx, any_alignment);
    // (the vector path of an upper solve is laid out for a virtual size)
    // one workgroup per row block, but no more than one per CU: workgroups
    // draw tickets until none is left (see the kernel)
    const unsigned row_blocks =
        trsv_num_blocks(n + (vec_ok && vec16 > 1
                                 ? trsv_virtual_pad<st_type>(upper, n)
                                 : index_type{0}));
    const bool persistent =
        num_cus > 0 && row_blocks > static_cast<unsigned>(num_cus);
    const dim3 grid(persistent ? static_cast<unsigned>(num_cus) : row_blocks,
                    1, 1);
#define ACCBLAS_TRSV_LAUNCH_RELAY(_upper, _unit, _relay)                       \
    if (vec_ok) {                                                              \
        if (persistent) {                                                      \
            kernel::trsv<waves, _upper, _unit, vec16, _relay, ArType, true>    \
                <<<grid, block, pad, stream>>>(mtx, x, helper, session);       \
        } else {                                                               \
            kernel::trsv<waves, _upper, _unit, vec16, _relay, ArType, false>   \
                <<<grid, block, pad, stream>>>(mtx, x, helper, session);       \
        }                                                                      \
    } else {                                                                   \
        if (persistent) {                                                      \
            kernel::trsv<waves, _upper, _unit, 1, _relay, ArType, true>        \
                <<<grid, block, pad, stream>>>(mtx, x, helper, session);       \
        } else {                                                               \
            kernel::trsv<waves, _upper, _unit, 1, _relay, ArType, false>       \
                <<<grid, block, pad, stream>>>(mtx, x, helper, session);       \
        }                                                                      \
    }
#define ACCBLAS_TRSV_LAUNCH(_upper, _unit)                                     \
    if (relay) {                                                               \
        ACCBLAS_TRSV_LAUNCH_RELAY(_upper, _unit, true)                         \
    } else {                                                                   \
        ACCBLAS_TRSV_LAUNCH_RELAY(_upper, _unit, false)                        \
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
#undef ACCBLAS_TRSV_LAUNCH_RELAY
    return hipGetLastError();
}


}  // namespace dev
}  // namespace accblas
