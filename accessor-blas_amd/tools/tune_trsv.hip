// Developer tool (not part of the library): times trsv kernel variants for
// Acc<fp64 arithmetic, fp32 storage>, lower / unit, on the current GPU.
// Usage: tune_trsv [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define ACCBLAS_TRSV_DIAG
#include "trsv_kernels.hpp"

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

struct entry {
    std::string name;
    std::function<void()> run;
};

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 65536;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
    const int stamp_entry = argc > 3 ? std::atoi(argv[3]) : 1;  // variant to stamp
    st_type *A, *x, *x0;
    std::uint32_t *helper;
    granule_word *granules;
    CHECK(hipMalloc(&A, size_t(n) * n * sizeof(st_type)));
    CHECK(hipMalloc(&x, n * sizeof(st_type)));
    CHECK(hipMalloc(&x0, n * sizeof(st_type)));
    CHECK(hipMalloc(&helper, 16));
    CHECK(hipMalloc(&granules, n * 16));
    CHECK(hipMemset(granules, 0, n * 16));
    trsv_state *state;
    CHECK(hipMalloc(&state, 64));
    const trsv_state initial{trsv_clean_tickets, 0u, trsv_first_epoch, 0u, 0u};
    CHECK(hipMemcpy(state, &initial, sizeof(initial), hipMemcpyHostToDevice));
    const trsv_session session{state, granules, index_type(n) * 2, 1u};
    fill<<<4096, 256>>>(A, size_t(n) * n, 1, 1.0f / std::sqrt(float(n)));
    fill<<<64, 256>>>(x0, n, 2, 1.0f);
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    c_range m(sz{{n, n}}, (const st_type *)A, st{{n}});
    range xv(sz{{n, 1}}, x, st{{1}});
    const unsigned blocks = unsigned((n + 63) / 64);

    auto make = [&](const char *name, auto kernel, int threads, size_t dyn_lds) {
        return entry{name, [=]() {
                         kernel<<<blocks, threads, dyn_lds>>>(m, xv, helper, session);
                     }};
    };
    std::vector<entry> entries = {
        // 8 worker waves + the loader wave = 576 threads
        make("direct 2/CU", kernel::trsv<8, false, true, 4, false, ar_type, false, c_range, range>, 512, 0),
        make("direct 1/CU", kernel::trsv<8, false, true, 4, false, ar_type, false, c_range, range>, 512, 40000),
        make("relay 1/CU", kernel::trsv<8, false, true, 4, true, ar_type, false, c_range, range>, 576, 40000),
        make("direct vec1 1/CU", kernel::trsv<8, false, true, 1, false, ar_type, false, c_range, range>, 512, 40000),
    };
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::vector<std::vector<float>> times(entries.size());
    for (int r = -1; r < rounds; ++r) {
        for (size_t i = 0; i < entries.size(); ++i) {
            CHECK(hipMemcpy(x, x0, n * sizeof(st_type), hipMemcpyDeviceToDevice));
            CHECK(hipEventRecord(start, nullptr));
            entries[i].run();
            CHECK(hipEventRecord(stop, nullptr));
            CHECK(hipEventSynchronize(stop));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) times[i].push_back(ms);
        }
    }
    const double bytes = sizeof(st_type) * (double(n) * (n - 1) / 2 + 2.0 * n);
    std::printf("n=%ld bytes=%.0f hops=%ld\n", n, bytes, (n + 63) / 64);
    for (size_t i = 0; i < entries.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        std::printf("%-16s min %9.1f us med %9.1f us -> %7.1f GB/s  %6.2f us/hop\n",
                    entries[i].name.c_str(), t.front() * 1e3, t[t.size() / 2] * 1e3,
                    bytes / (t.front() * 1e-3) / 1e9, t.front() * 1e3 / ((n + 63) / 64));
    }
    // phase stamps of the production shape (100 MHz clock -> us)
    {
        unsigned long long *diag;
        CHECK(hipMalloc(&diag, size_t(blocks) * 128));
        CHECK(hipMemset(diag, 0, size_t(blocks) * 128));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(trsv_diag_buffer), &diag, sizeof(diag)));
        CHECK(hipMemcpy(x, x0, n * sizeof(st_type), hipMemcpyDeviceToDevice));
        entries[stamp_entry].run();
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(size_t(blocks) * 16);
        CHECK(hipMemcpy(st.data(), diag, st.size() * 8, hipMemcpyDeviceToHost));
        const unsigned long long t0 = st[0];
        auto us = [&](unsigned b, int slot) { return (double)(long long)(st[b * 16 + slot] - t0) * 0.01; };
        std::printf("block: ticket  inv_done  w0_far  all_far  seen  published | hop  seen-prev_pub  pub-seen  allfar-prevprev_pub\n");
        for (unsigned b : {0u, 1u, 2u, 3u, 10u, 11u, 12u, 100u, 101u, 102u, 200u, 201u, 300u, 301u, 500u, 501u, 800u, 801u, 1000u, 1001u}) {
            if (b >= blocks) continue;
            std::printf("%5u: staged %6.2f M_ready %6.2f clk %4.0f MHz |", b, us(b, 10), us(b, 6),
                        b ? double(st[b * 16 + 12] - st[b * 16 + 11]) / (us(b, 5) - us(b, 4)) : 0.0);
            std::printf(" %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f | %6.2f %6.2f %6.2f %6.2f | seen->bar1 %5.2f matvec %5.2f bar2 %5.2f sum+pub %5.2f\n", us(b, 0), us(b, 1), us(b, 2),
                        us(b, 3), us(b, 4), us(b, 5), b ? us(b, 5) - us(b - 1, 5) : 0.0,
                        b ? us(b, 4) - us(b - 1, 5) : 0.0, us(b, 5) - us(b, 4),
                        b > 1 ? us(b, 3) - us(b - 2, 5) : 0.0, us(b, 7) - us(b, 4), us(b, 8) - us(b, 7),
                        us(b, 9) - us(b, 8), us(b, 5) - us(b, 9));
        }
        diag = nullptr;
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(trsv_diag_buffer), &diag, sizeof(diag)));
    }
    std::uint32_t h[2];
    CHECK(hipMemcpy(h, helper, 8, hipMemcpyDeviceToHost));
    std::printf("helper[0]=%08x\n", h[0]);
    return 0;
}
