// Developer tool (not part of the library): the dot at the shape one GPU of an
// 8-GPU run of config C4 sees - 2^25 fp16 pairs = 128 MiB, fp32 arithmetic -
// sourced from HBM: `sets` buffer pairs (default 8 = 1 GiB, four times the
// 256 MiB Infinity Cache) are swept in turn, so no launch finds its data in the
// cache. Per candidate shape: the minimum over 10 batches of the per-launch
// average of back-to-back launches over all sets (what bench.py reports), the
// minimum of event-timed SINGLE launches (each on the next set), and the same
// single launch on ONE buffer pair re-read (Infinity-Cache resident).
// Usage: tune_dot_chunk [log2 n = 25] [sets = 8] [st = f16|f32]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    std::function<void(int)> run;  // argument: buffer set
};

template <typename Ar, typename St>
void run_all(long n, int sets, dot_workspace &ws)
{
    using accessor = gko::acc::reduced_row_major<2, Ar, St>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    St *x, *y;
    Ar *res;
    CHECK(hipMalloc(&x, size_t(sets) * n * sizeof(St)));
    CHECK(hipMalloc(&y, size_t(sets) * n * sizeof(St)));
    CHECK(hipMalloc(&res, sizeof(Ar)));
    fill<<<4096, 256>>>(x, size_t(sets) * n, 1);
    fill<<<4096, 256>>>(y, size_t(sets) * n, 2);
    CHECK(hipDeviceSynchronize());
    dot_workspace *w = &ws;
    std::vector<entry> entries;
#define ADD(W, U, B, C)                                                          \
    entries.push_back({std::string("w" #W " u" #U " b" #B) + (C ? " contig" : " stride"), \
                       [=](int set) {                                            \
                           c_range xr(sz{{n, 1}}, (const St *)x + size_t(set) * n, st{{1}}); \
                           c_range yr(sz{{n, 1}}, (const St *)y + size_t(set) * n, st{{1}}); \
                           CHECK((launch_dot<Ar, dot_config<W, U, B, C>>(        \
                               *w, false, xr, yr, res, nullptr)));               \
                       }});
    ADD(8, 4, 2, false)  // production shape below 384 MiB
    ADD(8, 2, 2, false)
    ADD(8, 8, 2, false)
    ADD(8, 4, 1, false)
    ADD(8, 8, 1, false)
    ADD(8, 2, 4, false)
    ADD(8, 4, 4, false)
    ADD(4, 4, 4, false)
    ADD(4, 8, 2, false)
    ADD(4, 4, 2, false)
    ADD(16, 2, 1, false)
    ADD(16, 4, 1, false)
    ADD(8, 4, 2, true)
    ADD(8, 8, 1, true)   // production shape from 384 MiB on
    ADD(8, 4, 1, true)
    ADD(4, 8, 2, true)
    ADD(4, 4, 4, true)
#undef ADD
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    const double bytes = 2.0 * sizeof(St) * n;
    std::printf("n = %ld, %d buffer sets of 2 x %.0f MiB, us per launch (GB/s)\n", n, sets,
                bytes / 2 / 1048576.0);
    std::printf("%-20s %22s %22s %22s\n", "shape", "HBM back-to-back", "HBM single launch",
                "same buffers, single");
    std::vector<std::array<double, 3>> best(entries.size(), {1e30, 1e30, 1e30});
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t i0 = 0; i0 < entries.size(); ++i0) {
            const size_t i = pass == 0 ? i0 : entries.size() - 1 - i0;
            auto &e = entries[i];
            for (int s = 0; s < sets; ++s) e.run(s);  // warm-up
            CHECK(hipDeviceSynchronize());
            for (int batch = 0; batch < 10; ++batch) {
                CHECK(hipEventRecord(start, nullptr));
                for (int rep = 0; rep < 4; ++rep)
                    for (int s = 0; s < sets; ++s) e.run(s);
                CHECK(hipEventRecord(stop, nullptr));
                CHECK(hipEventSynchronize(stop));
                float ms;
                CHECK(hipEventElapsedTime(&ms, start, stop));
                best[i][0] = std::min(best[i][0], double(ms) * 1e3 / (4 * sets));
            }
            for (int rep = 0; rep < 3; ++rep) {
                for (int s = 0; s < sets; ++s) {
                    CHECK(hipEventRecord(start, nullptr));
                    e.run(s);
                    CHECK(hipEventRecord(stop, nullptr));
                    CHECK(hipEventSynchronize(stop));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, start, stop));
                    best[i][1] = std::min(best[i][1], double(ms) * 1e3);
                }
            }
            e.run(0);
            for (int rep = 0; rep < 12; ++rep) {
                CHECK(hipEventRecord(start, nullptr));
                e.run(0);
                CHECK(hipEventRecord(stop, nullptr));
                CHECK(hipEventSynchronize(stop));
                float ms;
                CHECK(hipEventElapsedTime(&ms, start, stop));
                best[i][2] = std::min(best[i][2], double(ms) * 1e3);
            }
        }
    }
    for (size_t i = 0; i < entries.size(); ++i) {
        std::printf("%-20s", entries[i].name.c_str());
        for (int k = 0; k < 3; ++k) {
            std::printf(" %12.2f (%6.0f)", best[i][k], bytes / best[i][k] / 1e3);
        }
        std::printf("\n");
    }
    CHECK(hipFree(x));
    CHECK(hipFree(y));
    CHECK(hipFree(res));
}

int main(int argc, char **argv)
{
    const long n = 1L << (argc > 1 ? std::atoi(argv[1]) : 25);
    const int sets = argc > 2 ? std::atoi(argv[2]) : 8;
    const bool f32 = argc > 3 && std::strcmp(argv[3], "f32") == 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    dot_workspace ws;
    char *buf;
    CHECK(hipMalloc(&buf, dot_slot_bytes + 16));
    CHECK(hipMemset(buf, 0, dot_slot_bytes + 16));
    ws.slots = (dot_slot_word *)buf;
    ws.scalar = buf + dot_slot_bytes;
    ws.num_cus = prop.multiProcessorCount;
    ws.serial = 1;
    if (f32) {
        run_all<double, float>(n, sets, ws);
    } else {
        run_all<float, _Float16>(n, sets, ws);
    }
    return 0;
}
