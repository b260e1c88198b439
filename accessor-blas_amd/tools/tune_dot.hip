// Developer tool (not part of the library): times dot kernel shapes on the
// current GPU, interleaved rounds in one process. Usage: tune_dot [rounds]
// or: tune_dot rounds n1 n2 ...  (the candidates for the size rule at these
// lengths, fp16 storage / fp32 arithmetic and fp32 storage / fp64 arithmetic)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "dot_kernels.hpp"

#define CHECK(call)                                                        \
    do {                                                                   \
        hipError_t e = (call);                                             \
        if (e != hipSuccess) {                                             \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,         \
                         hipGetErrorString(e));                            \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

using namespace accblas::dev;

template <typename T>
__global__ void fill(T *p, size_t n, unsigned seed)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = unsigned(i) * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = T((h & 0xffffff) / float(0x800000) - 1.0f);
    }
}

struct entry {
    std::string name;
    std::function<void()> run;
};

template <typename Ar, typename St>
struct problem {
    using accessor = gko::acc::reduced_row_major<2, Ar, St>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    St *x, *y;
    Ar *res;
    long n;
    dot_workspace ws;

    problem(long n_, const dot_workspace &ws_) : n(n_), ws(ws_)
    {
        CHECK(hipMalloc(&x, n * sizeof(St)));
        CHECK(hipMalloc(&y, n * sizeof(St)));
        CHECK(hipMalloc(&res, sizeof(Ar)));
        fill<<<2048, 256>>>(x, n, 1);
        fill<<<2048, 256>>>(y, n, 2);
    }
    template <typename Config>
    entry make(const char *name, bool atomic = false)
    {
        using sz = std::array<gko::acc::size_type, 2>;
        using st = std::array<gko::acc::size_type, 1>;
        c_range xr(sz{{n, 1}}, (const St *)x, st{{1}});
        c_range yr(sz{{n, 1}}, (const St *)y, st{{1}});
        auto *w = &ws;
        auto r = res;
        return {name, [=]() {
                    CHECK((launch_dot<Ar, Config>(*w, atomic, xr, yr, r, nullptr)));
                }};
    }
    template <typename F>
    void sweep(const char *title, int rounds, F add)
    {
        std::vector<entry> entries;
        add(entries);
        hipEvent_t start, stop;
        CHECK(hipEventCreate(&start));
        CHECK(hipEventCreate(&stop));
        std::vector<std::vector<float>> times(entries.size());
        for (auto &e : entries) e.run();
        CHECK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r) {
            for (size_t i = 0; i < entries.size(); ++i) {
                CHECK(hipEventRecord(start, nullptr));
                entries[i].run();
                CHECK(hipEventRecord(stop, nullptr));
                CHECK(hipEventSynchronize(stop));
                float ms;
                CHECK(hipEventElapsedTime(&ms, start, stop));
                times[i].push_back(ms);
            }
        }
        const double bytes = 2.0 * sizeof(St) * n;
        std::printf("%s n=%ld bytes=%.0f\n", title, n, bytes);
        for (size_t i = 0; i < entries.size(); ++i) {
            auto t = times[i];
            std::sort(t.begin(), t.end());
            std::printf("  %-26s min %8.2f us med %8.2f us -> %7.1f GB/s (min) %7.1f (med)\n",
                        entries[i].name.c_str(), t.front() * 1e3, t[t.size() / 2] * 1e3,
                        bytes / (t.front() * 1e-3) / 1e9, bytes / (t[t.size() / 2] * 1e-3) / 1e9);
        }
    }
};

template <typename P>
void add_all(P &p, std::vector<entry> &e)
{
    //                        waves unroll blocks/CU contiguous
    e.push_back(p.template make<dot_config<8, 4, 4, false>>("w8 u4 b4 stride"));
    e.push_back(p.template make<dot_config<8, 4, 4, true>>("w8 u4 b4 contig"));
    e.push_back(p.template make<dot_config<8, 4, 2, false>>("w8 u4 b2 stride"));
    e.push_back(p.template make<dot_config<8, 4, 2, true>>("w8 u4 b2 contig"));
    e.push_back(p.template make<dot_config<8, 2, 4, false>>("w8 u2 b4 stride"));
    e.push_back(p.template make<dot_config<8, 2, 4, true>>("w8 u2 b4 contig"));
    e.push_back(p.template make<dot_config<4, 4, 8, false>>("w4 u4 b8 stride"));
    e.push_back(p.template make<dot_config<4, 4, 8, true>>("w4 u4 b8 contig"));
    e.push_back(p.template make<dot_config<4, 4, 4, true>>("w4 u4 b4 contig"));
    e.push_back(p.template make<dot_config<16, 2, 2, false>>("w16 u2 b2 stride"));
    e.push_back(p.template make<dot_config<16, 2, 2, true>>("w16 u2 b2 contig"));
    e.push_back(p.template make<dot_config<8, 8, 2, true>>("w8 u8 b2 contig"));
    e.push_back(p.template make<dot_config<8, 4, 1, true>>("w8 u4 b1 contig"));
    e.push_back(p.template make<dot_config<8, 8, 1, true>>("w8 u8 b1 contig"));
    e.push_back(p.template make<dot_config<8, 6, 2, true>>("w8 u6 b2 contig"));
    e.push_back(p.template make<dot_config<8, 4, 3, true>>("w8 u4 b3 contig"));
    e.push_back(p.template make<dot_config<16, 4, 1, true>>("w16 u4 b1 contig"));
    e.push_back(p.template make<dot_config<16, 4, 2, true>>("w16 u4 b2 contig"));
    e.push_back(p.template make<dot_config<4, 8, 4, true>>("w4 u8 b4 contig"));
    e.push_back(p.template make<dot_config<8, 4, 4, false>>("w8 u4 b4 stride atomic", true));
}

template <typename P>
void add_rule_candidates(P &p, std::vector<entry> &e)
{
    e.push_back(p.template make<dot_config<8, 4, 2, false>>("w8 u4 b2 stride (small rule)"));
    e.push_back(p.template make<dot_config<8, 2, 4, false>>("w8 u2 b4 stride"));
    e.push_back(p.template make<dot_config<8, 8, 2, true>>("w8 u8 b2 contig"));
    e.push_back(p.template make<dot_config<8, 4, 2, true>>("w8 u4 b2 contig"));
    e.push_back(p.template make<dot_config<8, 8, 1, true>>("w8 u8 b1 contig (large rule)"));
    e.push_back(p.template make<dot_auto_config>("automatic"));
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 10;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    dot_workspace ws;
    char *buf;
    CHECK(hipMalloc(&buf, dot_slot_bytes + 16));
    CHECK(hipMemset(buf, 0, dot_slot_bytes + 16));
    ws.slots = (dot_slot_word *)buf;
    ws.scalar = buf + dot_slot_bytes;
    ws.serial = 1;
    ws.num_cus = prop.multiProcessorCount;
    if (argc > 2) {
        for (int a = 2; a < argc; ++a) {
            const long n = std::atol(argv[a]);
            {
                problem<float, _Float16> p(n, ws);
                p.sweep("Acc<f32,f16>", rounds, [&](std::vector<entry> &e) { add_rule_candidates(p, e); });
            }
            {
                problem<double, float> p(n / 2, ws);
                p.sweep("Acc<f64,f32>", rounds, [&](std::vector<entry> &e) { add_rule_candidates(p, e); });
            }
        }
        return 0;
    }
    {
        problem<double, float> p(1L << 28, ws);
        p.sweep("Acc<f64,f32>", rounds, [&](std::vector<entry> &e) { add_all(p, e); });
    }
    {
        problem<float, _Float16> p(1L << 28, ws);
        p.sweep("Acc<f32,f16>", rounds, [&](std::vector<entry> &e) { add_all(p, e); });
    }
    {
        problem<double, float> p(1L << 27, ws);
        p.sweep("Acc<f64,f32>", rounds, [&](std::vector<entry> &e) { add_all(p, e); });
    }
    {
        problem<float, _Float16> p(1L << 25, ws);
        p.sweep("Acc<f32,f16>", rounds, [&](std::vector<entry> &e) { add_all(p, e); });
    }
    {
        problem<double, double> p(1000000, ws);
        p.sweep("Acc<f64,f64>", rounds, [&](std::vector<entry> &e) { add_all(p, e); });
    }
    return 0;
}
