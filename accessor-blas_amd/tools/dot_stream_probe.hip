// Developer tool (not part of the library): what bounds the 1 GiB fp16 dot
// (config C4's per-launch workload)? Times the same two-stream reduction
// with different work distributions over the workgroups, plus a one-stream
// read of the same number of bytes as the ceiling.
// Usage: dot_stream_probe [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CHECK(call)                                                        \
    do {                                                                   \
        hipError_t e = (call);                                             \
        if (e != hipSuccess) {                                             \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,         \
                         hipGetErrorString(e));                            \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

using half8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ inline half8 load_nt(const half8 *p)
{
    return __builtin_nontemporal_load(p);
}

__global__ void fill(_Float16 *p, size_t n, unsigned seed)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = unsigned(i) * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = _Float16((h & 0xffffff) / float(0x800000) - 1.0f);
    }
}

__device__ inline float block_sum(float v, float *scratch)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < int(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

template <int U>
__device__ inline void fma_step(const half8 *x, const half8 *y, long base,
                                long stride, float (&acc)[U])
{
    half8 xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        xv[u] = load_nt(x + base + u * stride);
        yv[u] = load_nt(y + base + u * stride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[u] += float(xv[u][k]) * float(yv[u][k]);
}

// mode 0: contiguous slice per workgroup; mode 1: the same with the start
// rotated by `phase_step` iterations per workgroup; mode 2: chunks of one
// iteration handed out by an atomic counter (one ahead); mode 3: the whole
// grid strides together
template <int U, int Mode>
__global__ __launch_bounds__(512) void dot_probe(const half8 *x, const half8 *y,
                                                 long chunks, float *out,
                                                 unsigned *counter,
                                                 int phase_step)
{
    __shared__ float scratch[8];
    __shared__ unsigned next_item[2];
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    const long step = 512L * U;  // chunks per workgroup iteration
    const long iters_all = chunks / step;
    if (Mode == 0 || Mode == 1) {
        const long per = (iters_all + gridDim.x - 1) / gridDim.x;
        const long first = blockIdx.x * per;
        const long last = first + per < iters_all ? first + per : iters_all;
        const long count = last > first ? last - first : 0;
        long phase = Mode == 1 && count > 0
                         ? (long(blockIdx.x) * phase_step) % count
                         : 0;
        for (long i = 0; i < count; ++i) {
            long it = first + phase;
            phase = phase + 1 == count ? 0 : phase + 1;
            fma_step<U>(x, y, it * step + threadIdx.x, 512, acc);
        }
    } else if (Mode == 2) {
        if (threadIdx.x == 0) next_item[0] = atomicAdd(counter, 1u);
        __syncthreads();
        int slot = 0;
        while (true) {
            const unsigned it = next_item[slot];
            if (it >= iters_all) break;
            if (threadIdx.x == 0) next_item[slot ^ 1] = atomicAdd(counter, 1u);
            fma_step<U>(x, y, it * step + threadIdx.x, 512, acc);
            __syncthreads();
            slot ^= 1;
        }
    } else {
        const long grid_threads = long(gridDim.x) * 512;
        for (long c = blockIdx.x * 512L + threadIdx.x;
             c + (U - 1) * grid_threads < chunks; c += U * grid_threads)
            fma_step<U>(x, y, c, grid_threads, acc);
    }
    float v = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) v += acc[u];
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// one stream of the same number of bytes
template <int U>
__global__ __launch_bounds__(512) void read_probe(const half8 *x, long chunks,
                                                  float *out)
{
    __shared__ float scratch[8];
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    const long step = 512L * U;
    const long iters_all = chunks / step;
    const long per = (iters_all + gridDim.x - 1) / gridDim.x;
    const long first = blockIdx.x * per;
    const long last = first + per < iters_all ? first + per : iters_all;
    for (long it = first; it < last; ++it) {
        half8 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            xv[u] = load_nt(x + it * step + threadIdx.x + u * 512);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[u] += float(xv[u][k]);
    }
    float v = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) v += acc[u];
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

struct entry {
    std::string name;
    std::function<void()> run;
};

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 12;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long n = 1L << 28;
    const long chunks = n / 8;
    _Float16 *buf;
    // one allocation: x, then y at a chosen distance
    const size_t pad_max = 64 << 20;
    CHECK(hipMalloc(&buf, 2 * n * sizeof(_Float16) + pad_max));
    fill<<<4096, 256>>>(buf, n + n + pad_max / 2, 1);
    float *out;
    unsigned *counter;
    CHECK(hipMalloc(&out, 65536 * 4));
    CHECK(hipMalloc(&counter, 4));
    CHECK(hipDeviceSynchronize());
    const half8 *x = (const half8 *)buf;
    auto y_at = [&](size_t pad_bytes) {
        return (const half8 *)((const char *)buf + n * 2 + pad_bytes);
    };
    std::vector<entry> e;
    const half8 *y0 = y_at(0);
    e.push_back({"contig u8 b1 (library shape)", [=] {
                     dot_probe<8, 0><<<cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    for (int ps : {1, 3, 7, 13}) {
        e.push_back({"contig u8 b1 phase " + std::to_string(ps), [=] {
                         dot_probe<8, 1><<<cus, 512>>>(x, y0, chunks, out, counter, ps);
                     }});
    }
    e.push_back({"contig u4 b2 phase 3", [=] {
                     dot_probe<4, 1><<<2 * cus, 512>>>(x, y0, chunks, out, counter, 3);
                 }});
    e.push_back({"queue u8 b1", [=] {
                     CHECK(hipMemsetAsync(counter, 0, 4));
                     dot_probe<8, 2><<<cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    e.push_back({"queue u4 b2", [=] {
                     CHECK(hipMemsetAsync(counter, 0, 4));
                     dot_probe<4, 2><<<2 * cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    e.push_back({"queue u8 b2", [=] {
                     CHECK(hipMemsetAsync(counter, 0, 4));
                     dot_probe<8, 2><<<2 * cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    e.push_back({"stride u8 b1", [=] {
                     dot_probe<8, 3><<<cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    e.push_back({"stride u4 b2", [=] {
                     dot_probe<4, 3><<<2 * cus, 512>>>(x, y0, chunks, out, counter, 0);
                 }});
    for (size_t pad : {size_t(4096), size_t(65536), size_t(1 << 20) + 8192,
                       size_t(33 << 20) + 4096 * 5}) {
        const half8 *yp = y_at(pad);
        e.push_back({"contig u8 b1 y+" + std::to_string(pad), [=] {
                         dot_probe<8, 0><<<cus, 512>>>(x, yp, chunks, out, counter, 0);
                     }});
    }
    e.push_back({"one stream 1 GiB u8 b1", [=] {
                     read_probe<8><<<cus, 512>>>(x, 2 * chunks, out);
                 }});
    e.push_back({"one stream 1 GiB u16 b1", [=] {
                     read_probe<16><<<cus, 512>>>(x, 2 * chunks, out);
                 }});
    e.push_back({"one stream 1 GiB u8 b2", [=] {
                     read_probe<8><<<2 * cus, 512>>>(x, 2 * chunks, out);
                 }});

    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::vector<std::vector<float>> times(e.size());
    for (auto &it : e) it.run();
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = 0; i < e.size(); ++i) {
            CHECK(hipEventRecord(start, nullptr));
            e[i].run();
            CHECK(hipEventRecord(stop, nullptr));
            CHECK(hipEventSynchronize(stop));
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            times[i].push_back(ms);
        }
    }
    const double bytes = 4.0 * n;
    std::printf("fp16 two-stream reduction, n = 2^28 (%.0f bytes), %d CUs\n", bytes, cus);
    for (size_t i = 0; i < e.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        std::printf("  %-34s min %8.2f us med %8.2f us -> %7.1f GB/s (min) %7.1f (med)\n",
                    e[i].name.c_str(), t.front() * 1e3, t[t.size() / 2] * 1e3,
                    bytes / (t.front() * 1e-3) / 1e9,
                    bytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
    return 0;
}
