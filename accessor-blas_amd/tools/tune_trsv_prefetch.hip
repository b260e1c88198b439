// Developer experiment (not part of the library): does the 128-row TRSV's
// chain move faster when spare CUs pull the tiles of the next block columns
// into the Infinity Cache ahead of the front? The solve runs on one stream,
// a prefetcher kernel on another (it follows trsv_state::front and touches
// every cache line of block columns front+1 .. front+ahead of all later row
// blocks); both alternated with the solve alone, min / median of `rounds`.
// Acc<fp64, fp32>, lower / unit. Usage: tune_trsv_prefetch n rounds workgroups ahead...
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

// one workgroup of 256 threads per share of the row blocks; every thread
// touches two 128-byte lines of a 128 x 128 fp32 tile (512 lines)
__global__ __launch_bounds__(256) void prefetch_ahead(const st_type *A, long lda, int blocks,
                                                      const trsv_state *state, int ahead,
                                                      float *sink)
{
    int done = 2;  // block columns below this are prefetched (0 and 1 are needed at once)
    unsigned seen = 0;
    float sum = 0.0f;
    for (unsigned spins = 0; spins < (1u << 20); ++spins) {
        const unsigned front = __hip_atomic_load(&state->front, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        if (front < seen || front + 3 >= unsigned(blocks)) {
            break;  // the solve has re-armed its state, or nothing is left to fetch
        }
        seen = front;
        const int upto = std::min<int>(int(front) + 1 + ahead, blocks - 1);
        if (done < int(front) + 1) {
            done = int(front) + 1;  // fell behind: skip what the front has passed
        }
        if (done >= upto) {
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        for (int c = done; c < upto; ++c) {
            for (int b = c + 1 + int(blockIdx.x); b < blocks; b += int(gridDim.x)) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int line = int(threadIdx.x) + 256 * half;
                    const long row = long(b) * 128 + line / 4;
                    const long col = long(c) * 128 + (line % 4) * 32;
                    if (row < lda) {  // (square matrix, lda = n: the last row block may be ragged)
                        sum += A[row * lda + col];
                    }
                }
            }
        }
        done = upto;
    }
    if (sum == 12345.678f) {
        *sink = sum;
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: tune_trsv_prefetch n rounds workgroups ahead...\n");
        return 2;
    }
    const long n = std::atol(argv[1]);
    const int rounds = std::atoi(argv[2]);
    const int workgroups = std::atoi(argv[3]);
    st_type *A, *x, *x0;
    float *sink;
    std::uint32_t *helper;
    granule_word *granules;
    CHECK(hipMalloc(&A, size_t(n) * n * sizeof(st_type)));
    CHECK(hipMalloc(&x, n * sizeof(st_type)));
    CHECK(hipMalloc(&x0, n * sizeof(st_type)));
    CHECK(hipMalloc(&helper, 16));
    CHECK(hipMalloc(&sink, 16));
    const size_t ws_bytes = trsv_workspace_bytes<double>(n);
    CHECK(hipMalloc(&granules, ws_bytes));
    CHECK(hipMemset(granules, 0, ws_bytes));
    trsv_state *state;
    CHECK(hipMalloc(&state, 64));
    const trsv_state initial{trsv_clean_tickets, 0u, trsv_first_epoch, 0u, 0u};
    CHECK(hipMemcpy(state, &initial, sizeof(initial), hipMemcpyHostToDevice));
    trsv_session session{state, granules, index_type(ws_bytes / 8), 1u};
    fill<<<4096, 256>>>(A, size_t(n) * n, 1, 1.0f / std::sqrt(float(n)));
    fill<<<64, 256>>>(x0, n, 2, 1.0f);
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    c_range m(sz{{n, n}}, (const st_type *)A, st{{n}});
    range xv(sz{{n, 1}}, x, st{{1}});
    const int blocks = int((n + 127) / 128);
    hipStream_t solve_stream, fetch_stream;
    CHECK(hipStreamCreate(&solve_stream));
    CHECK(hipStreamCreate(&fetch_stream));
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::vector<int> aheads{0};
    for (int a = 4; a < argc; ++a) aheads.push_back(std::atoi(argv[a]));
    std::vector<std::vector<float>> times(aheads.size());
    std::vector<double> checks(aheads.size());
    std::vector<st_type> host(n);
    for (int r = -1; r < rounds; ++r) {
        for (size_t i = 0; i < aheads.size(); ++i) {
            CHECK(hipMemcpy(x, x0, n * sizeof(st_type), hipMemcpyDeviceToDevice));
            CHECK(hipDeviceSynchronize());
            session.serial = trsv_next_serial();
            CHECK(hipEventRecord(start, solve_stream));
            if (aheads[i] > 0) {
                prefetch_ahead<<<workgroups, 256, 0, fetch_stream>>>(A, n, blocks, state, aheads[i], sink);
            }
            CHECK(launch_trsv_wide<ar_type>(false, true, m, xv, helper, session, solve_stream, 256));
            CHECK(hipEventRecord(stop, solve_stream));
            CHECK(hipEventSynchronize(stop));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) times[i].push_back(ms);
            CHECK(hipMemcpy(host.data(), x, n * sizeof(st_type), hipMemcpyDeviceToHost));
            double s = 0;
            for (long k = 0; k < n; ++k) s += std::fabs(double(host[k]));
            checks[i] = s;
        }
    }
    std::uint32_t h[2];
    CHECK(hipMemcpy(h, helper, 8, hipMemcpyDeviceToHost));
    std::printf("n=%ld (%d row blocks), %d prefetching workgroups, helper=%08x\n", n, blocks, workgroups, h[0]);
    for (size_t i = 0; i < aheads.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        std::printf("  ahead %2d: min %8.1f med %8.1f us   sum|x| %.9e\n", aheads[i], t.front() * 1e3,
                    t[t.size() / 2] * 1e3, checks[i]);
    }
    return 0;
}
