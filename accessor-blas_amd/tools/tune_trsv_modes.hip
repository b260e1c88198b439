// Developer tool (not part of the library): why does the 64-row relay kernel
// with 8-byte storage have two run-time modes between n = 20000 and 24000
// (profiles/r04_trsv_plain_vs_acc.txt)? Runs `rounds` stamped solves of
// trsv<8, Upper, Unit, 2, Relay> for plain views and accessor ranges in turn
// and prints, for the fastest and the slowest solve of each, where the chain
// was when (per 16 row blocks: ticket drawn, far sweep finished, x_{b-1} seen,
// published) and how many blocks were NOT ready when their turn came.
// Usage: tune_trsv_modes [n] [rounds] [lda] [relay 0|1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
using T = double;
using accessor = gko::acc::reduced_row_major<2, T, T>;
using range = gko::acc::range<accessor>;
using c_range = gko::acc::range<accessor::const_accessor>;

__global__ void fill(T *p, size_t n, unsigned seed, double scale)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = unsigned(i) * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffffff) / double(0x800000) - 1.0) * scale;
    }
}

struct run_record {
    float us;
    std::vector<unsigned long long> stamps;
};

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 24000;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 24;
    const long lda = argc > 3 ? std::atol(argv[3]) : n;
    const bool relay = argc > 4 ? std::atoi(argv[4]) != 0 : true;
    T *A, *x, *x0;
    std::uint32_t *helper;
    granule_word *granules;
    CHECK(hipMalloc(&A, size_t(n) * lda * sizeof(T)));
    CHECK(hipMalloc(&x, n * sizeof(T)));
    CHECK(hipMalloc(&x0, n * sizeof(T)));
    CHECK(hipMalloc(&helper, 16));
    const size_t ws_bytes = trsv_workspace_bytes<T>(n);
    CHECK(hipMalloc(&granules, ws_bytes));
    CHECK(hipMemset(granules, 0, ws_bytes));
    trsv_state *state;
    CHECK(hipMalloc(&state, 4096));
    const trsv_state initial{trsv_clean_tickets, 0u, trsv_first_epoch, 0u, 0u};
    CHECK(hipMemcpy(state, &initial, sizeof(initial), hipMemcpyHostToDevice));
    fill<<<4096, 256>>>(A, size_t(n) * lda, 1, 1.0 / std::sqrt(double(n)));
    fill<<<64, 256>>>(x0, n, 2, 1.0);
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    const c_range m_acc(sz{{n, n}}, (const T *)A, st{{lda}});
    const range x_acc(sz{{n, 1}}, x, st{{1}});
    const plain_view<const T> m_plain{A, {n, n}, lda};
    const plain_view<T> x_plain{x, {n, 1}, 1};
    const long padded = n + trsv_virtual_pad<T>(true, n);
    const unsigned blocks = unsigned((padded + 63) / 64);
    const unsigned threads = relay ? 576 : 512;
    const unsigned dyn = relay ? trsv_single_block_pad<T, 8, true>()
                               : trsv_single_block_pad<T, 8, false>();
    unsigned long long *diag;
    CHECK(hipMalloc(&diag, size_t(blocks) * 128));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(trsv_diag_buffer), &diag, sizeof(diag)));
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::uint32_t serial = 1;
    std::vector<run_record> records[2];
    for (int r = -2; r < rounds; ++r) {
        for (int which = 0; which < 2; ++which) {
            CHECK(hipMemcpy(x, x0, n * sizeof(T), hipMemcpyDeviceToDevice));
            CHECK(hipMemset(diag, 0, size_t(blocks) * 128));
            CHECK(hipDeviceSynchronize());
            const trsv_session session{state, granules,
                                       index_type(ws_bytes / 8), ++serial};
            CHECK(hipEventRecord(start, nullptr));
            if (which == 0) {
                if (relay) {
                    kernel::trsv<8, true, true, 2, true, T>
                        <<<blocks, threads, dyn>>>(m_plain, x_plain, helper, session);
                } else {
                    kernel::trsv<8, true, true, 2, false, T>
                        <<<blocks, threads, dyn>>>(m_plain, x_plain, helper, session);
                }
            } else {
                if (relay) {
                    kernel::trsv<8, true, true, 2, true, T>
                        <<<blocks, threads, dyn>>>(m_acc, x_acc, helper, session);
                } else {
                    kernel::trsv<8, true, true, 2, false, T>
                        <<<blocks, threads, dyn>>>(m_acc, x_acc, helper, session);
                }
            }
            CHECK(hipEventRecord(stop, nullptr));
            CHECK(hipEventSynchronize(stop));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) {
                run_record rec{ms * 1e3f, std::vector<unsigned long long>(size_t(blocks) * 16)};
                CHECK(hipMemcpy(rec.stamps.data(), diag, rec.stamps.size() * 8,
                                hipMemcpyDeviceToHost));
                records[which].push_back(std::move(rec));
            }
        }
    }
    std::printf("n=%ld lda=%ld blocks=%u relay=%d (stamped solves, us)\n", n, lda, blocks, int(relay));
    for (int which = 0; which < 2; ++which) {
        auto &recs = records[which];
        std::vector<float> t;
        for (auto &r : recs) t.push_back(r.us);
        std::printf("%s:", which == 0 ? "plain" : "acc  ");
        for (float v : t) std::printf(" %.0f", v);
        std::printf("\n");
        auto fastest = std::min_element(recs.begin(), recs.end(),
                                        [](auto &a, auto &b) { return a.us < b.us; });
        auto slowest = std::max_element(recs.begin(), recs.end(),
                                        [](auto &a, auto &b) { return a.us < b.us; });
        for (auto *rec : {&*fastest, &*slowest}) {
            const auto &s = rec->stamps;
            unsigned long long t0 = ~0ull;
            for (unsigned b = 0; b < blocks; ++b) {
                if (s[b * 16] != 0) t0 = std::min(t0, s[b * 16]);
            }
            auto us = [&](unsigned b, int slot) {
                return (double)(long long)(s[b * 16 + slot] - t0) * 0.01;
            };
            int late = 0, late_after_256 = 0;
            double worst_wait = 0;
            for (unsigned b = 2; b < blocks; ++b) {
                // not ready when its turn came: far sums after x_{b-1} was published
                if (us(b, 3) > us(b - 1, 5)) {
                    ++late;
                    if (b >= 256) ++late_after_256;
                    worst_wait = std::max(worst_wait, us(b, 3) - us(b - 1, 5));
                }
            }
            std::printf("  %s solve %.0f us: blocks not ready at their turn %d (%d of them >= 256), "
                        "worst lateness %.1f us, last published at %.1f\n",
                        rec == &*fastest ? "fastest" : "slowest", rec->us, late,
                        late_after_256, worst_wait, us(blocks - 1, 5));
            std::printf("    block: ticket  staged  M_ready  far_done  seen  published | hop(avg of 16)  far_done-prev_pub | first wait in the sweep: at, panel, front then\n");
            for (unsigned b = 0; b < blocks; b += 16) {
                const double hop16 = b >= 16 ? (us(b, 5) - us(b - 16, 5)) / 16 : 0.0;
                // where the chain's front was when this block first had to wait for x
                unsigned front = 0;
                const double waited = s[b * 16 + 13] ? us(b, 13) : -1.0;
                while (waited >= 0 && front + 1 < blocks && us(front, 5) < waited) ++front;
                std::printf("    %5u: %7.1f %7.1f %7.1f %8.1f %8.1f %8.1f | %5.2f  %7.1f | %8.1f %5llu %5u\n", b,
                            us(b, 0), us(b, 10), us(b, 6), us(b, 3), us(b, 4), us(b, 5), hop16,
                            b ? us(b, 3) - us(b - 1, 5) : 0.0, waited, s[b * 16 + 14], front);
            }
        }
    }
    std::uint32_t h[2];
    CHECK(hipMemcpy(h, helper, 8, hipMemcpyDeviceToHost));
    std::printf("helper[0]=%08x\n", h[0]);
    return 0;
}
