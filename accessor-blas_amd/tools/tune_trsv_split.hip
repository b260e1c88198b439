// Developer tool (not part of the library): the 128-row TRSV kernel under
// several row-block splits, alternated in ONE process (boxes and moments
// differ by +-3 %, more than most of these choices). Acc<fp64, fp32>, lower /
// unit. Usage: tune_trsv_split n rounds first:parts:tail_blocks:tail_parts:tail_deferred[:last_blocks:last_parts] ...
// (first = 0: the library's rule for everything)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "trsv_wide_kernels.hpp"

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
using ar_type = double;
using st_type = float;
using accessor = gko::acc::reduced_row_major<2, ar_type, st_type>;
using range = gko::acc::range<accessor>;
using c_range = gko::acc::range<accessor::const_accessor>;

__global__ void fill(st_type *p, size_t n, unsigned seed, float scale)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = unsigned(i) * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffffff) / float(0x800000) - 1.0f) * scale;
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: tune_trsv_split n rounds first:parts:tail_blocks:tail_parts:tail_deferred[:last_blocks:last_parts] ...\n");
        return 2;
    }
    const long n = std::atol(argv[1]);
    const int rounds = std::atoi(argv[2]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = int((n + 127) / 128);
    std::vector<std::string> names;
    std::vector<trsv_wide_split> splits;
    for (int a = 3; a < argc; ++a) {
        int first, parts, tail_blocks, tail_parts, tail_deferred, last_blocks = 0, last_parts = 1;
        if (std::sscanf(argv[a], "%d:%d:%d:%d:%d:%d:%d", &first, &parts, &tail_blocks, &tail_parts,
                        &tail_deferred, &last_blocks, &last_parts) < 5) {
            std::fprintf(stderr, "bad config %s\n", argv[a]);
            return 2;
        }
        trsv_wide_split split = trsv_wide_split_for(n, prop.multiProcessorCount);
        if (first > 0) {
            if (first < 98 || parts < 2 || parts > trsv_wide_max_parts || tail_parts < 2 ||
                tail_parts > trsv_wide_max_parts || tail_deferred < 1 || tail_deferred > 48 ||
                last_parts < 1 || last_parts > trsv_wide_max_parts || last_blocks < 0) {
                std::fprintf(stderr, "config out of range %s\n", argv[a]);
                return 2;
            }
            if (blocks > first) {
                tail_blocks = std::min(tail_blocks, blocks - first);
                last_blocks = std::min(last_blocks, tail_blocks);
                split = trsv_wide_split{first, parts, blocks - tail_blocks, tail_parts, tail_deferred,
                                        blocks - last_blocks, last_parts};
            } else {
                split = trsv_wide_split{blocks, 1, blocks, 1, 48, blocks, 1};
            }
        }
        names.push_back(argv[a]);
        splits.push_back(split);
    }
    st_type *A, *x, *x0;
    std::uint32_t *helper;
    granule_word *granules;
    CHECK(hipMalloc(&A, size_t(n) * n * sizeof(st_type)));
    CHECK(hipMalloc(&x, n * sizeof(st_type)));
    CHECK(hipMalloc(&x0, n * sizeof(st_type)));
    CHECK(hipMalloc(&helper, 16));
    const size_t ws_bytes = trsv_workspace_bytes<double>(n);
    CHECK(hipMalloc(&granules, ws_bytes));
    CHECK(hipMemset(granules, 0, ws_bytes));
    trsv_state *state;
    CHECK(hipMalloc(&state, 64));
    const trsv_state initial{trsv_clean_tickets, 0u, trsv_first_epoch, 0u, 0u};
    CHECK(hipMemcpy(state, &initial, sizeof(initial), hipMemcpyHostToDevice));
    const trsv_session session{state, granules, index_type(ws_bytes / 8), 1u};
    fill<<<4096, 256>>>(A, size_t(n) * n, 1, 1.0f / std::sqrt(float(n)));
    fill<<<64, 256>>>(x0, n, 2, 1.0f);
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    c_range m(sz{{n, n}}, (const st_type *)A, st{{n}});
    range xv(sz{{n, 1}}, x, st{{1}});
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::vector<std::vector<float>> times(splits.size());
    std::vector<double> checksum(splits.size());
    std::vector<st_type> host(n);
    for (int r = -1; r < rounds; ++r) {
        for (size_t i = 0; i < splits.size(); ++i) {
            CHECK(hipMemcpy(x, x0, n * sizeof(st_type), hipMemcpyDeviceToDevice));
            CHECK(hipEventRecord(start, nullptr));
            // (the CU count as the library passes it: row blocks below it fold at once)
            CHECK(launch_trsv_wide_split<ar_type>(false, true, m, xv, helper, session, nullptr, splits[i],
                                                  prop.multiProcessorCount));
            CHECK(hipEventRecord(stop, nullptr));
            CHECK(hipEventSynchronize(stop));
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) times[i].push_back(ms);
            if (r == 0) {
                CHECK(hipMemcpy(host.data(), x, n * sizeof(st_type), hipMemcpyDeviceToHost));
                double s = 0;
                for (long k = 0; k < n; ++k) s += std::fabs(double(host[k]));
                checksum[i] = s;
            }
        }
    }
    const double bytes = sizeof(st_type) * (double(n) * (n - 1) / 2 + 2.0 * n);
    std::uint32_t h[2];
    CHECK(hipMemcpy(h, helper, 8, hipMemcpyDeviceToHost));
    std::printf("n=%ld (%d row blocks) rounds=%d helper=%08x\n", n, blocks, rounds, h[0]);
    for (size_t i = 0; i < splits.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        std::printf("  %-24s first %3d parts %2d tail from %3d parts %2d keep %2d last from %3d parts %2d | min %8.1f med %8.1f us | %6.1f GB/s (min) | sum|x| %.9e\n",
                    names[i].c_str(), splits[i].first_block, splits[i].parts, splits[i].tail_block,
                    splits[i].tail_parts, splits[i].tail_deferred, splits[i].last_block, splits[i].last_parts,
                    t.front() * 1e3, t[t.size() / 2] * 1e3,
                    bytes / (t.front() * 1e-3) / 1e9, checksum[i]);
    }
    return 0;
}
