// Developer microbenchmark (round 5): what does the TRSV hop's "seen after
// publication" latency consist of when the POLLING CU is itself streaming
// matrix tiles? Two workgroups (512 threads, one per CU, on different XCDs)
// play ping-pong with a 1 KB block of 128 tagged granules {serial, value},
// as the 128-row solve does. Wave 0 of each polls / publishes; waves 1..7
// keep a deep queue of 16-byte non-temporal loads in flight (the tile ring),
// and optionally every other CU of the chip streams too.
//
// Poll paths compared:
//   vec      global_load_dwordx4 sc1 of the whole block (the library's poll)
//   vec+prio the same with s_setprio 3 on the polling wave
//   vec ld-nt / st-nt / nt-nt: the non-temporal hint on the poll, the
//            publishing store, or both
//   sK       s_load_dwordx16 glc of K 64-byte lines of the block (scalar data
//            cache path: does not queue in the CU's vector memory pipeline)
//   sK+inv   the same behind s_dcache_inv
//   s1>vec   scalar poll of the block's last line, then ONE vector load of
//            the block (what a two-stage hand-over would cost)
//   s8w      ALL eight waves poll two lines each (both loads issued, one wait)
//            and meet through an LDS counter: the whole block through scalar
//            registers, as a kernel would have to do it (do the scalar cache's
//            misses overlap?); v8w = the same with 16-lane vector loads
// Memory kinds for the granules: ordinary hipMalloc and uncached
// (hipDeviceMallocUncached), because a scalar load has no sc1 bit: whether it
// sees a remote XCD's write-through store is part of the question.
// The chain's time is taken inside the kernel (s_memrealtime, 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/hop_paths_probe tools/hop_paths_probe.hip
// (stands by itself: no library header). Results: profiles/r05_trsv_scalar_poll.txt
// Usage: hop_paths_probe [hops]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                          \
    do {                                                                     \
        hipError_t e = (call);                                               \
        if (e != hipSuccess) {                                               \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,           \
                         hipGetErrorString(e));                              \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

typedef float vec4 __attribute__((ext_vector_type(4)));
typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
typedef unsigned uvec16 __attribute__((ext_vector_type(16)));

enum path { VEC = 0, VEC_PRIO = 1, S1 = 2, S4 = 3, S1_INV = 4, S1_THEN_VEC = 5, S2 = 6, S8W = 7, V8W = 8, VEC_NT = 9, VEC_STNT = 10, VEC_NTNT = 11 };

__device__ __forceinline__ int xcc_id()
{
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 0xf;
}

__device__ __forceinline__ unsigned long long realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

// one 64-byte line through the scalar path, bypassing the scalar cache
__device__ __forceinline__ uvec16 scalar_line(const unsigned *p)
{
    uvec16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(p) : "memory");
    return v;
}

// tags sit in the odd dwords of a line: {value, serial} x 8
__device__ __forceinline__ bool line_has(const uvec16 v, unsigned serial)
{
    return v[1] == serial && v[3] == serial && v[5] == serial && v[7] == serial &&
           v[9] == serial && v[11] == serial && v[13] == serial && v[15] == serial;
}

__device__ __forceinline__ uvec4 vector_block(const unsigned *block, int lane, bool nt = false)
{
    uvec4 v;
    const unsigned *p = block + lane * 4;
    if (nt) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

// state: [0..15] arrivals per XCD, [16] failure, [17] done, [32..33] XCC ids,
// [34..35] chain time in 10 ns ticks (lo, hi)
template <int Path>
__global__ __launch_bounds__(512) void chain(unsigned *g, int hops, unsigned *state,
                                             const vec4 *buf, size_t count, int own_load,
                                             int chip_load, float *sink)
{
    extern __shared__ unsigned lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int xcc = xcc_id();
    if (threadIdx.x == 0) {
        lds[0] = 0;
        lds[2] = 0;
        lds[1] = atomicAdd(&state[xcc], 1u);
    }
    __syncthreads();
    const int arrival = (int)lds[1];
    const int me = (arrival == 0 && xcc < 2) ? xcc : -1;
    float acc = 0;
    if (me < 0) {
        // the rest of the chip: stream until the chain is over (bounded)
        if (chip_load) {
            size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) % count;
            for (int it = 0; it < (1 << 14); ++it) {
                for (int k = 0; k < 8; ++k) {
                    const vec4 v = __builtin_nontemporal_load(buf + i);
                    acc += v.x + v.w;
                    i += (size_t)gridDim.x * 512;
                    if (i >= count) i -= count;
                }
                unsigned done;
                const unsigned *dp = state + 17;
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=v"(done) : "v"(dp) : "memory");
                if (done) break;
            }
        }
        if (acc == 12345.678f) *sink = acc;
        return;
    }
    if (Path == S8W || Path == V8W) {
        // every wave waits for its two lines of each block this participant
        // consumes, then counts itself in; wave 0 publishes when all eight have
        unsigned long long t0 = 0;
        bool failed = false;
        if (wave == 0 && lane == 0) state[32 + me] = xcc;
        for (int h = me; h < hops && !failed; h += 2) {
            if (h > 0) {
                const unsigned long long mine_bits =
                    (unsigned long long)(g + (size_t)(h - 1) * 256 + wave * 32);
                const unsigned *mine = (const unsigned *)(
                    ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mine_bits >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)mine_bits));
                const unsigned serial = (unsigned)h;
                unsigned spins = 0;
                while (true) {
                    bool seen;
                    if (Path == S8W) {
                        uvec16 a, b;
                        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\t"
                                     "s_load_dwordx16 %1, %2, 0x40 glc\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&s"(a), "=&s"(b) : "s"(mine) : "memory");
                        seen = line_has(a, serial) && line_has(b, serial);
                    } else {
                        unsigned long long v = 0;
                        const unsigned *p = mine + (lane & 15) * 2;
                        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=v"(v) : "v"(p) : "memory");
                        seen = __all((unsigned)(v >> 32) == serial);
                    }
                    if (seen) break;
                    if (++spins > (1u << 18)) {
                        failed = true;
                        break;
                    }
                }
                if (lane == 0) atomicAdd(&lds[2], 1u);  // LDS
                if (wave != 0) continue;
                // wave 0: all eight waves of this hop counted in (8 per consumed hop)
                const unsigned want = 8u * (unsigned)((h - me) / 2 + (me == 0 ? 0 : 1));
                unsigned polls = 0;
                while (*(volatile unsigned *)&lds[2] < want && ++polls < (1u << 22)) {
                }
            } else {
                if (wave != 0) continue;
                t0 = realtime();
            }
            if (wave == 0) {
                uvec4 out;
                out[0] = (unsigned)lane;
                out[1] = (unsigned)h + 1;
                out[2] = (unsigned)lane + 64;
                out[3] = (unsigned)h + 1;
                unsigned *dst = g + (size_t)h * 256 + lane * 4;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(out) : "memory");
            }
        }
        if (wave == 0) {
            if (failed && lane == 0) state[16] = 1;
            if (me == (hops - 1) % 2 || failed) {
                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
                if (lane == 0) {
                    unsigned one = 1;
                    unsigned *dp = state + 17;
                    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(dp), "v"(one) : "memory");
                }
            }
            if (me == 0) {
                const unsigned long long t1 = realtime();
                if (lane == 0) {
                    state[34] = (unsigned)(t1 - t0);
                    state[35] = (unsigned)((t1 - t0) >> 32);
                }
            }
        }
        return;
    }
    if (wave != 0) {
        // the participants' own tile ring: 8 x 16 bytes per lane in flight
        if (own_load) {
            size_t i = ((size_t)(me * 4096 + wave) * 64 + lane) % count;
            for (int it = 0; it < (1 << 16); ++it) {
                vec4 r[8];
                for (int k = 0; k < 8; ++k) {
                    r[k] = __builtin_nontemporal_load(buf + i);
                    i += 8191 * 64;
                    if (i >= count) i -= count;
                }
                for (int k = 0; k < 8; ++k) acc += r[k].x + r[k].w;
                if (*(volatile unsigned *)&lds[0]) break;
            }
        }
        if (acc == 12345.678f) *sink = acc;
        return;
    }
    if (Path == VEC_PRIO) __builtin_amdgcn_s_setprio(3);
    if (lane == 0) state[32 + me] = xcc;
    unsigned long long t0 = 0;
    bool failed = false;
    for (int h = me; h < hops && !failed; h += 2) {
        if (h > 0) {
            const unsigned *block = g + (size_t)(h - 1) * 256;
            const unsigned serial = (unsigned)h;  // hop h - 1 carries serial h
            unsigned spins = 0;
            while (true) {
                bool seen;
                if (Path == VEC || Path == VEC_PRIO || Path == VEC_NT || Path == VEC_STNT ||
                    Path == VEC_NTNT) {
                    const uvec4 v = vector_block(block, lane, Path == VEC_NT || Path == VEC_NTNT);
                    seen = __all(v[1] == serial && v[3] == serial);
                } else if (Path == S1 || Path == S1_INV || Path == S1_THEN_VEC) {
                    if (Path == S1_INV) asm volatile("s_dcache_inv" : : : "memory");
                    seen = line_has(scalar_line(block + 240), serial);
                } else if (Path == S2) {
                    seen = line_has(scalar_line(block), serial) &&
                           line_has(scalar_line(block + 240), serial);
                } else {
                    seen = line_has(scalar_line(block), serial) &&
                           line_has(scalar_line(block + 80), serial) &&
                           line_has(scalar_line(block + 160), serial) &&
                           line_has(scalar_line(block + 240), serial);
                }
                if (seen) break;
                if (++spins > (1u << 18)) {
                    failed = true;
                    break;
                }
            }
            if (failed) break;
            if (Path == S1_THEN_VEC) {
                const uvec4 v = vector_block(block, lane);
                if (!__all(v[1] == serial && v[3] == serial)) acc += 1;  // torn: counted below
            }
        } else {
            t0 = realtime();
        }
        uvec4 out;
        out[0] = (unsigned)lane;
        out[1] = (unsigned)h + 1;
        out[2] = (unsigned)lane + 64;
        out[3] = (unsigned)h + 1;
        unsigned *dst = g + (size_t)h * 256 + lane * 4;
        if (Path == VEC_STNT || Path == VEC_NTNT) {
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(dst), "v"(out) : "memory");
        } else {
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(out) : "memory");
        }
    }
    if (failed && lane == 0) state[16] = 1;
    if (acc != 0 && lane == 0) atomicAdd(&state[18], 1u);
    if (me == (hops - 1) % 2 || failed) {
        // the participant that published the last hop closes the run
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        if (lane == 0) {
            unsigned one = 1;
            unsigned *dp = state + 17;
            asm volatile("global_store_dword %0, %1, off sc1" : : "v"(dp), "v"(one) : "memory");
        }
    }
    if (me == 0) {
        // participant 0 saw (or published) the chain's whole length but for one hop
        const unsigned long long t1 = realtime();
        if (lane == 0) {
            state[34] = (unsigned)(t1 - t0);
            state[35] = (unsigned)((t1 - t0) >> 32);
        }
    }
    lds[0] = 1;
}

template <int Path>
void run(const char *name, unsigned *g, const char *kind, int hops, int own, int chip,
         unsigned *state, const vec4 *buf, size_t count, float *sink)
{
    double best = 1e30;
    unsigned host_state[40] = {};
    bool bad = false;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipMemset(g, 0, (size_t)hops * 1024));
        CHECK(hipMemset(state, 0, 40 * 4));
        CHECK(hipDeviceSynchronize());
        chain<Path><<<256, 512, 128 * 1024>>>(g, hops, state, buf, count, own, chip, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost));
        if (host_state[16]) bad = true;
        const double us = (double)host_state[34] * 0.01;
        if (us < best) best = us;
    }
    std::printf("%-10s %-9s own=%d chip=%d xcc(%u,%u) %s torn=%u %8.3f us/hop\n", name, kind, own,
                chip, host_state[32], host_state[33], bad ? "NOT VISIBLE" : "ok         ",
                host_state[18], best / hops);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int hops = argc > 1 ? std::atoi(argv[1]) : 1000;
    unsigned *g_plain, *g_unc, *state;
    float *sink;
    vec4 *buf;
    const size_t count = (size_t{1} << 30) / 16 * 2;  // 2 GiB
    CHECK(hipMalloc(&g_plain, (size_t)hops * 1024));
    CHECK(hipExtMallocWithFlags((void **)&g_unc, (size_t)hops * 1024, hipDeviceMallocUncached));
    CHECK(hipMalloc(&state, 40 * 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&buf, count * 16));
    CHECK(hipMemset(buf, 0, count * 16));
    CHECK(hipFuncSetAttribute((const void *)chain<VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<VEC_PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S1_INV>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S1_THEN_VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<VEC_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<VEC_STNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<VEC_NTNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<S8W>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void *)chain<V8W>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int mode = 0; mode < 4; ++mode) {
        // idle / the two CUs' other waves stream / the whole chip streams /
        // the chip streams but the two CUs do not (the solve's tail: the
        // block's sweep is over, the rest of the chip is in its own)
        const int own = mode == 1 || mode == 2, chip = mode >= 2;
        for (int kind = 0; kind < 2; ++kind) {
            unsigned *g = kind ? g_unc : g_plain;
            const char *kn = kind ? "uncached" : "hipMalloc";
#define RUN(P, text) run<P>(text, g, kn, hops, own, chip, state, buf, count, sink)
            RUN(VEC, "vec");
            RUN(VEC_PRIO, "vec+prio");
            RUN(VEC_NT, "vec ld-nt");
            RUN(VEC_STNT, "vec st-nt");
            RUN(VEC_NTNT, "vec nt-nt");
            RUN(S1, "s1");
            RUN(S1_INV, "s1+inv");
            RUN(S2, "s2");
            RUN(S4, "s4");
            RUN(S1_THEN_VEC, "s1>vec");
            if (!own) {
                RUN(S8W, "s8w");
                RUN(V8W, "v8w");
            }
#undef RUN
        }
    }
    return 0;
}
