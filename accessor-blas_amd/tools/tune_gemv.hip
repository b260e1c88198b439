// Developer tool (not part of the library): times gemv_vec shapes for
// Acc<fp64 arithmetic, fp32 storage> at n x n on the current GPU, interleaved
// rounds in one process (guide §5.4 rule 24). Usage: tune_gemv [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "gemv_kernels.hpp"

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

__global__ void fill(st_type *p, size_t n, unsigned seed)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = unsigned(i) * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0xffffff) / float(0x800000) - 1.0f;
    }
}

struct entry {
    std::string name;
    std::function<void()> run;
};
// Experimental variant (north_star: "LDS staging of x-vector tiles"): every
// wave owns R rows, all waves of the workgroup sweep the same columns, and the
// x tile is staged ONCE per workgroup in LDS, already widened to the
// arithmetic type, double-buffered. Requires cols % TileCols == 0.
template <int Waves, int R, int U, int TileCols, bool NT>
__global__ __launch_bounds__(Waves * 64) void gemv_lds(double alpha, c_range mtx, c_range x,
                                                      double beta, range res)
{
    constexpr int Vec = 4;
    __shared__ double xs[2][TileCols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = mtx.length(0), cols = mtx.length(1);
    const long row0 = (long(blockIdx.x) * Waves + wave) * R;
    const float *a = mtx->get_const_storage();
    const float *xp = x->get_const_storage();
    const long lda = mtx->stride(0);
    const float *row_ptr[R];
    for (int j = 0; j < R; ++j) row_ptr[j] = a + (row0 + j < rows ? row0 + j : rows - 1) * lda;
    double acc[R] = {};
    auto stage = [&](long tile, int buf) {
        for (int i = threadIdx.x * Vec; i < TileCols; i += Waves * 64 * Vec) {
            const auto v = load_packed<Vec, false>(xp + tile * TileCols + i);
            for (int k = 0; k < Vec; ++k) xs[buf][i + k] = double(v[k]);
        }
    };
    const long tiles = cols / TileCols;
    stage(0, 0);
    for (long t = 0; t < tiles; ++t) {
        __syncthreads();
        if (t + 1 < tiles) stage(t + 1, (t + 1) & 1);
        const double *xt = xs[t & 1];
        for (int c0 = 0; c0 < TileCols; c0 += 64 * Vec * U) {
            typename packed<float, Vec>::type av[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j)
                    av[j][u] = load_packed<Vec, NT>(row_ptr[j] + t * TileCols + c0 + u * 64 * Vec + lane * Vec);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < Vec; ++k) {
                    const double xk = xt[c0 + u * 64 * Vec + lane * Vec + k];
#pragma unroll
                    for (int j = 0; j < R; ++j) acc[j] += double(av[j][u][k]) * xk;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double total = wave_reduce_sum(acc[j]);
        if (lane == 0 && row0 + j < rows) res(row0 + j, 0) = alpha * total + beta * res(row0 + j, 0);
    }
}

template <int Waves, int R, int U, int TileCols, bool NT>
entry make_lds(const char *name, c_range m, c_range x, range r)
{
    return {name, [=]() {
                const unsigned grid = unsigned((m.length(0) + Waves * R - 1) / (Waves * R));
                gemv_lds<Waves, R, U, TileCols, NT><<<grid, Waves * 64>>>(1.0, m, x, 1.0, r);
            }};
}


template <typename Config>
entry make(const char *name, c_range m, c_range x, range r)
{
    return {name, [=]() { CHECK(launch_gemv<Config>(1.0, m, x, 1.0, r, nullptr)); }};
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 32768;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 10;
    st_type *A, *x, *res;
    CHECK(hipMalloc(&A, size_t(n) * n * sizeof(st_type)));
    CHECK(hipMalloc(&x, n * sizeof(st_type)));
    CHECK(hipMalloc(&res, n * sizeof(st_type)));
    fill<<<4096, 256>>>(A, size_t(n) * n, 1);
    fill<<<64, 256>>>(x, n, 2);
    CHECK(hipMemset(res, 0, n * sizeof(st_type)));
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    c_range m(sz{{n, n}}, (const st_type *)A, st{{n}});
    c_range xv(sz{{n, 1}}, (const st_type *)x, st{{1}});
    range rv(sz{{n, 1}}, res, st{{1}});

    std::vector<entry> entries;
#define ADD(W, WPR, R, U) \
    entries.push_back(make<gemv_config<W, WPR, R, U, true>>("w" #W " wpr" #WPR " r" #R " u" #U " nt", m, xv, rv));
    ADD(8, 8, 4, 2)
    ADD(8, 8, 1, 8)
    ADD(8, 8, 1, 16)
    ADD(8, 8, 2, 4)
    ADD(8, 8, 2, 8)
    ADD(8, 8, 4, 1)
    ADD(8, 8, 4, 3)
    ADD(8, 8, 8, 2)
    ADD(8, 4, 4, 2)
    ADD(8, 4, 2, 4)
    ADD(8, 4, 8, 1)
    ADD(8, 2, 4, 2)
    ADD(8, 2, 8, 1)
    ADD(8, 1, 4, 2)
    ADD(8, 1, 8, 1)
    ADD(4, 4, 4, 2)
    ADD(4, 4, 2, 4)
    ADD(4, 4, 8, 2)
    ADD(4, 2, 4, 2)
    ADD(4, 1, 8, 1)
    ADD(16, 16, 2, 2)
    ADD(16, 8, 4, 2)
    ADD(16, 4, 4, 2)
    ADD(8, 8, 4, 2)
#undef ADD
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::vector<std::vector<float>> times(entries.size());
    for (auto &e : entries) {  // warm-up
        e.run();
    }
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
    const double bytes = sizeof(st_type) * (double(n) * n + 3.0 * n);
    std::printf("n=%ld bytes=%.0f\n", n, bytes);
    for (size_t i = 0; i < entries.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        const double mn = t.front(), med = t[t.size() / 2];
        std::printf("%-22s min %8.3f us  med %8.3f us  -> %7.1f GB/s (min) %7.1f GB/s (med)\n",
                    entries[i].name.c_str(), mn * 1e3, med * 1e3,
                    bytes / (mn * 1e-3) / 1e9, bytes / (med * 1e-3) / 1e9);
    }
    return 0;
}
