// Developer microbenchmark: cost of one granule hand-off hop between two
// workgroups (64 lanes publish 64 granules; the other side polls them), the
// primitive the TRSV chain is made of. Usage: pingpong [hops]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(call) do { hipError_t e = (call); if (e != hipSuccess) { \
    std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); std::exit(1);} } while (0)

using word = unsigned long long;

__device__ __forceinline__ word ld(const word *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(word *p, word v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NB blocks take turns: block (h % NB) publishes hop h after seeing hop h-1.
// Mode 0: poll with s_sleep(1); 1: poll without sleep; 2: 4 waves poll staggered,
// first to see claims via LDS; 3: work of ~W FMA steps between see and publish
template <int Mode>
__global__ __launch_bounds__(256) void chain(word *g, int hops, int nb, int work, double *sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    __shared__ int claimed_hop;
    if (threadIdx.x == 0) claimed_hop = -1;
    __syncthreads();
    if (Mode != 2 && wave != 0) return;
    double acc = lane;
    for (int h = blockIdx.x; h < hops; h += nb) {
        if (h > 0) {
            const word *src = g + (size_t)(h - 1) * 64 + lane;
            if (Mode == 2) {
                for (int d = 0; d < wave; ++d) __builtin_amdgcn_s_sleep(6);
            }
            unsigned spins = 0;
            bool mine = true;
            while (true) {
                const word v = ld(src);
                if (__all((v >> 32) == 1)) break;
                if (Mode == 2 && *(volatile int *)&claimed_hop >= h) { mine = false; break; }
                if (++spins > (1u << 22)) break;
                if (Mode == 0) __builtin_amdgcn_s_sleep(1);
            }
            if (Mode == 2) {
                if (mine) {
                    int prev = 0;
                    if (lane == 0) prev = atomicMax(&claimed_hop, h);
                    prev = __builtin_amdgcn_readfirstlane(prev);
                    if (prev >= h) mine = false;
                }
                if (!mine) continue;
            }
        }
        for (int w = 0; w < work; ++w) acc = acc * 1.0000001 + 0.5;
        st(g + (size_t)h * 64 + lane, (word{1} << 32) | (unsigned)(int)acc);
    }
    if (acc == 12345.678) *sink = acc;
}

template <int Mode>
void run(const char *name, word *g, int hops, int nb, int work, double *sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipMemset(g, 0, (size_t)hops * 64 * 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        chain<Mode><<<nb, 256>>>(g, hops, nb, work, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    std::printf("%-28s nb=%3d work=%4d: %8.1f us total, %6.3f us/hop\n", name, nb, work,
                best * 1e3, best * 1e3 / hops);
}

// effective shader clock while a kernel with `active` busy waves per block runs:
// ratio of s_memtime (shader cycles) to s_memrealtime (100 MHz)
__global__ void clock_probe(unsigned long long *out, int iters, double *sink)
{
    double acc = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = acc * 1.0000001 + 0.5;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (acc == 12345.678) *sink = acc;
}

int main(int argc, char **argv)
{
    {
        unsigned long long *out; double *sink2;
        CHECK(hipMalloc(&out, 4096 * 16)); CHECK(hipMalloc(&sink2, 8));
        for (int blocks : {1, 16, 256, 2048}) {
            for (int rep = 0; rep < 2; ++rep) {
                clock_probe<<<blocks, 64>>>(out, 200000, sink2);
                CHECK(hipDeviceSynchronize());
            }
            unsigned long long h[2];
            CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            std::printf("clock probe blocks=%4d: %llu shader cycles in %llu x10ns -> %.0f MHz, %.1f cycles per dependent f64 fma iteration\n",
                        blocks, h[0], h[1], h[0] / (h[1] * 0.01), double(h[0]) / 200000);
        }
    }
    const int hops = argc > 1 ? std::atoi(argv[1]) : 2000;
    word *g; double *sink;
    CHECK(hipMalloc(&g, (size_t)hops * 64 * 8));
    CHECK(hipMalloc(&sink, 8));
    for (int nb : {2, 8, 64, 256}) {
        run<0>("poll+sleep", g, hops, nb, 0, sink);
        run<1>("poll tight", g, hops, nb, 0, sink);
        run<2>("4 staggered waves", g, hops, nb, 0, sink);
    }
    run<1>("poll tight", g, hops, 64, 100, sink);
    run<1>("poll tight", g, hops, 64, 400, sink);
    return 0;
}
