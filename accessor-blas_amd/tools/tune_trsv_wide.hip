// Developer tool (not part of the library): times the 128-row TRSV kernel
// against the 64-row one for Acc<fp64 arithmetic, fp32 storage>, lower / unit,
// and prints a per-block timeline from in-register stamps.
// Usage: tune_trsv_wide [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define ACCBLAS_TRSV_WIDE_DIAG
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
#ifndef TUNE_AR_TYPE
#define TUNE_AR_TYPE double
#endif
using ar_type = TUNE_AR_TYPE;
#ifndef TUNE_ST_TYPE
#define TUNE_ST_TYPE float
#endif
using st_type = TUNE_ST_TYPE;
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
        p[i] = static_cast<st_type>(((h & 0xffffff) / float(0x800000) - 1.0f) * scale);
    }
}

int main(int argc, char **argv)
{
    // (UPLO=upper in the environment: the mirrored solve)
    const char *uplo_env = std::getenv("UPLO");
    const bool upper_solve = uplo_env != nullptr && std::string(uplo_env) == "upper";
    const long n = argc > 1 ? std::atol(argv[1]) : 65536;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
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
    const unsigned blocks = unsigned((n + 127) / 128);
    // argv[5]: first:parts:tail_blocks:tail_parts:tail_deferred[:last_blocks:last_parts]
    // as tune_trsv_split takes it (default: the library's rule)
    trsv_wide_split split = trsv_wide_split_for(n, 256);
    if (argc > 5) {
        int first, parts, tail_blocks, tail_parts, tail_deferred, last_blocks = 0, last_parts = 1;
        if (std::sscanf(argv[5], "%d:%d:%d:%d:%d:%d:%d", &first, &parts, &tail_blocks, &tail_parts,
                        &tail_deferred, &last_blocks, &last_parts) < 5) {
            std::fprintf(stderr, "bad split %s\n", argv[5]);
            return 2;
        }
        if (first > 0 && int(blocks) > first) {
            tail_blocks = std::min(tail_blocks, int(blocks) - first);
            last_blocks = std::min(last_blocks, tail_blocks);
            split = trsv_wide_split{first, parts, int(blocks) - tail_blocks, tail_parts, tail_deferred,
                                    int(blocks) - last_blocks, last_parts};
        }
    }
    std::printf("split: first %d parts %d tail from %d parts %d keep %d last from %d parts %d\n", split.first_block,
                split.parts, split.tail_block, split.tail_parts, split.tail_deferred, split.last_block,
                split.last_parts);
    struct entry { std::string name; std::function<void()> run; };
    std::vector<entry> entries = {
        {"rows64", [&]() { CHECK(launch_trsv<ar_type>(upper_solve, true, m, xv, helper, session, nullptr, 1)); }},
        {"rows128", [&]() { CHECK(launch_trsv_wide_split<ar_type>(upper_solve, true, m, xv, helper, session, nullptr, split, 256)); }},
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
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) times[i].push_back(ms);
        }
    }
    const double bytes = sizeof(st_type) * (double(n) * (n - 1) / 2 + 2.0 * n);
    std::printf("n=%ld bytes=%.0f\n", n, bytes);
    for (size_t i = 0; i < entries.size(); ++i) {
        auto t = times[i];
        std::sort(t.begin(), t.end());
        std::printf("%-10s min %9.1f us med %9.1f us -> %7.1f GB/s  %6.2f us per 64 rows\n",
                    entries[i].name.c_str(), t.front() * 1e3, t[t.size() / 2] * 1e3,
                    bytes / (t.front() * 1e-3) / 1e9, t.front() * 1e3 / ((n + 63) / 64));
    }
    // timeline of the 128-row kernel (100 MHz stamps -> us since block 0 started)
    unsigned long long *diag;
    CHECK(hipMalloc(&diag, size_t(blocks) * 8 * 20 * 8));
    CHECK(hipMemset(diag, 0, size_t(blocks) * 8 * 20 * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(trsv_wide_diag_buffer), &diag, sizeof(diag)));
    unsigned long long *records;
    CHECK(hipMalloc(&records, size_t(blocks) * trsv_wide_max_parts * 8 * 8));
    CHECK(hipMemset(records, 0, size_t(blocks) * trsv_wide_max_parts * 8 * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(trsv_wide_diag_workgroups), &records, sizeof(records)));
    CHECK(hipMemcpy(x, x0, n * sizeof(st_type), hipMemcpyDeviceToDevice));
    entries[1].run();
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> s(size_t(blocks) * 8 * 20);
    CHECK(hipMemcpy(s.data(), diag, s.size() * 8, hipMemcpyDeviceToHost));
    const unsigned long long t0 = s[0];
    auto us = [&](unsigned b, int w, int slot) {
        return (double)(long long)(s[(size_t(b) * 8 + w) * 20 + slot] - t0) * 0.01;
    };
    for (unsigned b : {0u, 1u, 8u, 50u, 100u}) {
        if (b >= blocks) continue;
        std::printf("prologue %3u (w0 / w1 / w7): start %.2f | ticket %.2f %.2f %.2f | staged %.2f %.2f %.2f | "
                    "inverted %.2f %.2f %.2f | X10 %.2f %.2f %.2f | W or sweep start %.2f %.2f %.2f\n",
                    b, us(b, 0, 0), us(b, 0, 12), us(b, 1, 12), us(b, 7, 12), us(b, 0, 13), us(b, 1, 13),
                    us(b, 7, 13), us(b, 0, 14), us(b, 1, 14), us(b, 7, 14), us(b, 0, 15), us(b, 1, 15),
                    us(b, 7, 15), us(b, 0, 1), us(b, 1, 1), us(b, 7, 1));
    }
    std::printf("columns: | at_last_shared(w0, w7) observed |\n");
    std::printf("block: start  W_ready | w0: far_last reduced | all_far  c_ready(w0,w7) | seen  handed  "
                "mv_done(w0,w7)  pub(w0,w7) | hop  seen-prev_pub(max)  pub-seen  c_ready-prevprev_pub\n");
    auto pub = [&](unsigned b) { double m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, us(b, w, 9)); return m; };
    for (unsigned b : {0u, 1u, 2u, 3u, 10u, 11u, 30u, 31u, 60u, 61u, 100u, 101u, 150u, 151u, 186u, 187u, 250u, 251u, 400u, 401u, 510u, 511u}) {
        if (b >= blocks) continue;
        std::printf("%4u: %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f %8.2f | %8.2f  %8.2f %8.2f | %8.2f %8.2f  %8.2f %8.2f  %8.2f %8.2f | %5.2f %5.2f %5.2f %6.2f\n",
                    b, us(b, 0, 0), us(b, 7, 1), us(b, 0, 10), us(b, 7, 10), us(b, 0, 11), us(b, 0, 2), us(b, 0, 3), us(b, 0, 4), us(b, 0, 5), us(b, 7, 5),
                    b ? us(b, 0, 6) : 0.0, b ? us(b, 0, 7) : 0.0, b ? us(b, 0, 8) : 0.0, b ? us(b, 7, 8) : 0.0,
                    us(b, 0, 9), us(b, 7, 9),
                    b ? pub(b) - pub(b - 1) : 0.0, b ? us(b, 0, 6) - pub(b - 1) : 0.0,
                    b ? pub(b) - us(b, 0, 6) : 0.0, b > 1 ? std::max(us(b, 0, 5), us(b, 7, 5)) - pub(b - 2) : 0.0);
    }
    // where the bytes flow: every workgroup's columns, spread over the time it
    // caught up in bulk and then placed at the moments the front handed them over
    {
        std::vector<unsigned long long> rec(size_t(blocks) * trsv_wide_max_parts * 8);
        CHECK(hipMemcpy(rec.data(), records, rec.size() * 8, hipMemcpyDeviceToHost));
        auto t_of = [&](unsigned long long v) { return (double)(long long)(v - t0) * 0.01; };
        std::vector<double> pubt(blocks);
        for (unsigned b = 0; b < blocks; ++b) pubt[b] = pub(b);
        const double bucket = argc > 3 ? std::atof(argv[3]) : 50.0;
        const double total = pubt[blocks - 1];
        const int buckets = int(total / bucket) + 2;
        std::vector<double> tiles(buckets, 0.0), resident(buckets, 0.0), following(buckets, 0.0);
        auto add = [&](double t, double amount) {
            int i = int(t / bucket);
            i = std::min(std::max(i, 0), buckets - 1);
            tiles[i] += amount;
        };
        auto add_span = [&](double a, double b, double amount, std::vector<double> &into, bool rate) {
            if (b <= a) { b = a + 0.01; }
            for (int i = std::max(0, int(a / bucket)); i < buckets && i * bucket < b; ++i) {
                const double lo = std::max(a, i * bucket), hi = std::min(b, (i + 1) * bucket);
                if (hi > lo) into[i] += rate ? amount * (hi - lo) / bucket : amount * (hi - lo) / (b - a);
            }
        };
        for (unsigned b = 0; b < blocks; ++b) {
            for (int p = 0; p < trsv_wide_max_parts; ++p) {
                const unsigned long long *r = &rec[(size_t(b) * trsv_wide_max_parts + p) * 8];
                if (r[7] != 1) continue;
                const double start = t_of(r[0]), sweep0 = t_of(r[1]), end = t_of(r[4]);
                const long c0 = long(r[5]), c1 = long(r[6]);
                add_span(start, end, 1.0, resident, true);
                if (r[2] == 0) {
                    add_span(sweep0, end, double(std::max(0L, c1 - c0)), tiles, false);
                } else {
                    const double tf = t_of(r[2]);
                    const long cf = long(r[3]);
                    add_span(sweep0, tf, double(std::max(0L, cf - c0)), tiles, false);
                    for (long c = cf; c < c1; ++c) add(std::max(pubt[c], tf), 1.0);
                    add_span(tf, end, 1.0, following, true);
                }
            }
            add(pubt[b], b > 0 ? 1.5 : 0.5);  // folded panel + triangle
        }
        std::printf("bytes over time (%.0f us buckets): start us | TB/s | front (row blocks published) | "
                    "workgroups resident | of them at the front\n", bucket);
        double sum = 0;
        for (int i = 0; i < buckets; ++i) {
            unsigned front = 0;
            while (front < blocks && pubt[front] <= (i + 1) * bucket) ++front;
            sum += tiles[i];
            std::printf("  %7.0f | %5.2f | %4u | %6.1f | %6.1f\n", i * bucket,
                        tiles[i] * 128.0 * 128.0 * sizeof(st_type) / (bucket * 1e-6) / 1e12, front,
                        resident[i], following[i]);
        }
        std::printf("  accounted: %.3f GB of %.3f GB\n", sum * 128.0 * 128.0 * sizeof(st_type) / 1e9, bytes / 1e9);
    }
    // the last hops one by one: owner start, ready for the front (c), hop
    std::printf("last row blocks: block | owner start | c ready | x_{b-1} seen - published | hop\n");
    for (unsigned b = blocks > 96 ? blocks - 96 : 1; b < blocks; ++b) {
        std::printf("  %4u | %8.2f | %8.2f | %6.2f | %6.2f\n", b, us(b, 0, 0),
                    std::max(us(b, 0, 5), us(b, 7, 5)), us(b, 0, 6) - pub(b - 1), pub(b) - pub(b - 1));
    }
    // the hop's own wait, probe by probe: when the predecessor's last wave
    // issued its publishing store, when the probe that saw it was issued, when
    // it was seen; probes issued while waiting and the mean time per probe
    std::printf("hop wait: block | c ready -> seen us | probes | us per probe | prev pub -> issue of the "
                "seeing probe | its round trip | pub spread of prev (min..max over waves)\n");
    for (unsigned b = 8; b < blocks; b += (blocks > 64 ? blocks / 24 : 1)) {
        const double c_ready = us(b, 0, 5), seen = us(b, 0, 6);
        const double probes = double(s[(size_t(b) * 8 + 0) * 20 + 18]);
        const unsigned long long issue_raw = s[(size_t(b) * 8 + 0) * 20 + 19];
        const double issue = issue_raw ? (double)(long long)(issue_raw - t0) * 0.01 : c_ready;
        double pmin = 1e30, pmax = 0;
        for (int w = 0; w < 8; ++w) { pmin = std::min(pmin, us(b - 1, w, 9)); pmax = std::max(pmax, us(b - 1, w, 9)); }
        std::printf("  %4u | %6.2f | %4.0f | %5.2f | %6.2f | %5.2f | %5.2f .. %5.2f\n", b, seen - c_ready, probes,
                    probes > 0 ? (seen - c_ready) / (probes + 1) : 0.0, issue - pmax, seen - issue,
                    pmin - pub(b - 1), pmax - pub(b - 1));
    }
    // every stamp of every block and wave (us since block 0 started; raw 0 = not
    // taken) for offline analysis: argv[4] = file name
    if (argc > 4) {
        std::FILE *f = std::fopen(argv[4], "w");
        if (f != nullptr) {
            std::fprintf(f, "block,wave");
            for (int k = 0; k < 20; ++k) std::fprintf(f, ",s%d", k);
            std::fprintf(f, "\n");
            for (unsigned b = 0; b < blocks; ++b) {
                for (int w = 0; w < 8; ++w) {
                    std::fprintf(f, "%u,%d", b, w);
                    for (int k = 0; k < 20; ++k) {
                        const unsigned long long raw = s[(size_t(b) * 8 + w) * 20 + k];
                        if (k == 18) {
                            std::fprintf(f, ",%llu", raw);
                        } else {
                            std::fprintf(f, ",%.2f", raw ? (double)(long long)(raw - t0) * 0.01 : -1.0);
                        }
                    }
                    std::fprintf(f, "\n");
                }
            }
            std::fclose(f);
        }
    }
    std::uint32_t h[2];
    CHECK(hipMemcpy(h, helper, 8, hipMemcpyDeviceToHost));
    std::printf("helper[0]=%08x\n", h[0]);
    return 0;
}
