// Developer microbenchmark: one granule hand-off hop between two workgroups
// that sit on the SAME XCD versus on DIFFERENT XCDs, polled with loads of
// different cache scopes, idle and beside a streaming kernel. Answers whether
// a TRSV chain that stays inside one XCD's L2 for a run of hops would be
// faster than the agent-scope (sc1) hand-off the library uses.
// Usage: xcd_hop [hops]
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

using word = unsigned long long;

enum scope { SC1 = 0, SC0 = 1, SC0_SC1 = 2, NONE = 3 };

template <int Scope>
__device__ __forceinline__ word poll_load(const word *p)
{
    word v;
    if (Scope == SC1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else if (Scope == SC0) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else if (Scope == SC0_SC1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                     : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

template <int Scope>
__device__ __forceinline__ void publish(word *p, word v)
{
    if (Scope == SC1) {
        asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    } else if (Scope == SC0) {
        asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
    } else if (Scope == SC0_SC1) {
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
    }
}

__device__ __forceinline__ int xcc_id()
{
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 0xf;
}

// Two participants take turns (hop h is published by participant h % 2 after
// it has seen hop h - 1). same_xcd: the first two workgroups that land on XCD
// 0; otherwise the first workgroup on XCD 0 and the first on XCD 1. `state`:
// [0..15] arrival counters per XCD, [16] failure flag, [32..33] the XCC ids of
// the participants as observed.
template <int LoadScope, int StoreScope>
__global__ __launch_bounds__(64) void chain(word *g, int hops, int same_xcd,
                                            unsigned *state)
{
    const int lane = threadIdx.x;
    const int xcc = xcc_id();
    int arrival = 0;
    if (lane == 0) {
        arrival = atomicAdd(&state[xcc], 1u);
    }
    arrival = __builtin_amdgcn_readfirstlane(arrival);
    int me = -1;
    if (same_xcd) {
        if (xcc == 0 && arrival < 2) me = arrival;
    } else {
        if (arrival == 0 && xcc < 2) me = xcc;
    }
    if (me < 0) return;
    if (lane == 0) state[32 + me] = xcc;
    for (int h = me; h < hops; h += 2) {
        if (h > 0) {
            const word *src = g + (size_t)(h - 1) * 64 + lane;
            unsigned spins = 0;
            while (true) {
                const word v = poll_load<LoadScope>(src);
                if (__all((v >> 32) == 1)) break;
                if (++spins > (1u << 20)) {
                    if (lane == 0) state[16] = 1;  // never became visible
                    return;
                }
            }
        }
        publish<StoreScope>(g + (size_t)h * 64 + lane, (word{1} << 32) | (unsigned)h);
    }
}

// background: every CU streams `buf` with non-temporal 16-byte loads
typedef float vec4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void streamer(const vec4 *buf, size_t count,
                                                int rounds, float *sink)
{
    float acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count;
             i += (size_t)gridDim.x * 256) {
            const vec4 v = __builtin_nontemporal_load(buf + i);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int LoadScope, int StoreScope>
void run(const char *name, word *g, int hops, bool same, bool loaded,
         unsigned *state, const vec4 *buf, size_t count, float *sink,
         hipStream_t s_chain, hipStream_t s_load)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    unsigned host_state[40] = {};
    for (int r = 0; r < 3; ++r) {
        CHECK(hipMemset(g, 0, (size_t)hops * 64 * 8));
        CHECK(hipMemset(state, 0, 40 * 4));
        CHECK(hipDeviceSynchronize());
        if (loaded) {
            streamer<<<256, 256, 0, s_load>>>(buf, count, 6, sink);
        }
        CHECK(hipEventRecord(a, s_chain));
        chain<LoadScope, StoreScope><<<256, 64, 0, s_chain>>>(g, hops, same ? 1 : 0, state);
        CHECK(hipEventRecord(b, s_chain));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost));
        best = ms < best ? ms : best;
    }
    std::printf("%-34s %-5s %-7s xcc(%u,%u) %s %8.3f us/hop\n", name,
                same ? "same" : "cross", loaded ? "loaded" : "idle",
                host_state[32], host_state[33],
                host_state[16] ? "NOT VISIBLE" : "ok        ", best * 1e3 / hops);
}

int main(int argc, char **argv)
{
    const int hops = argc > 1 ? std::atoi(argv[1]) : 2000;
    word *g;
    unsigned *state;
    float *sink;
    vec4 *buf;
    const size_t count = (size_t{1} << 30) / 16 * 2;  // 2 GiB
    CHECK(hipMalloc(&g, (size_t)hops * 64 * 8));
    CHECK(hipMalloc(&state, 40 * 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&buf, count * 16));
    CHECK(hipMemset(buf, 0, count * 16));
    hipStream_t s_chain, s_load;
    CHECK(hipStreamCreate(&s_chain));
    CHECK(hipStreamCreate(&s_load));
    for (int loaded = 0; loaded < 2; ++loaded) {
        for (int same = 1; same >= 0; --same) {
#define RUN(L, S, text) \
    run<L, S>(text, g, hops, same, loaded, state, buf, count, sink, s_chain, s_load)
            RUN(SC1, SC1, "load sc1 / store sc1");
            RUN(SC0, SC1, "load sc0 / store sc1");
            RUN(SC0, SC0, "load sc0 / store sc0");
            RUN(SC0_SC1, SC0_SC1, "load sc0 sc1 / store sc0 sc1");
            RUN(NONE, SC1, "load plain / store sc1");
#undef RUN
        }
    }
    return 0;
}
