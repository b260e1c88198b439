// Developer tool (not part of the library): the 64-row relay kernel with 8-byte
// storage, plain views and accessor ranges (identical instruction streams, see
// profiles/r04_trsv_plain_vs_acc.txt), alternated solve by solve WITHOUT stamps.
// Built several times with -DACCBLAS_TRSV_ENTRY_NOPS=k (k s_nop at the kernel's
// entry shift all of its code by 4k bytes): does the placement of the code
// decide which of the two is the faster one?
// Usage: tune_trsv_align [n] [reps] [lda]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

#ifndef ACCBLAS_TRSV_ENTRY_NOPS
#define ACCBLAS_TRSV_ENTRY_NOPS 0
#endif

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 24000;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 30;
    const long lda = argc > 3 ? std::atol(argv[3]) : n;
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
    const unsigned dyn = trsv_single_block_pad<T, 8, true>();
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    std::uint32_t serial = 1;
    std::vector<float> times[2];
    for (int r = -2; r < reps; ++r) {
        for (int which = 0; which < 2; ++which) {
            CHECK(hipMemcpy(x, x0, n * sizeof(T), hipMemcpyDeviceToDevice));
            CHECK(hipDeviceSynchronize());
            const trsv_session session{state, granules,
                                       index_type(ws_bytes / 8), ++serial};
            CHECK(hipEventRecord(start, nullptr));
            if (which == 0) {
                kernel::trsv<8, true, true, 2, true, T>
                    <<<blocks, 576, dyn>>>(m_plain, x_plain, helper, session);
            } else {
                kernel::trsv<8, true, true, 2, true, T>
                    <<<blocks, 576, dyn>>>(m_acc, x_acc, helper, session);
            }
            CHECK(hipEventRecord(stop, nullptr));
            CHECK(hipEventSynchronize(stop));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            if (r >= 0) times[which].push_back(ms * 1e3f);
        }
    }
    hipFuncAttributes attr_plain, attr_acc;
    CHECK(hipFuncGetAttributes(
        &attr_plain, reinterpret_cast<const void *>(
                         &kernel::trsv<8, true, true, 2, true, T, false, plain_view<const T>,
                                       plain_view<T>>)));
    CHECK(hipFuncGetAttributes(
        &attr_acc,
        reinterpret_cast<const void *>(&kernel::trsv<8, true, true, 2, true, T, false, c_range, range>)));
    std::printf("entry nops %d, n=%ld lda=%ld: registers plain %d acc %d\n",
                ACCBLAS_TRSV_ENTRY_NOPS, n, lda, attr_plain.numRegs, attr_acc.numRegs);
    for (int which = 0; which < 2; ++which) {
        auto t = times[which];
        std::sort(t.begin(), t.end());
        std::printf("  %s min %6.1f  p25 %6.1f  med %6.1f  p75 %6.1f  max %6.1f |", which == 0 ? "plain" : "acc  ",
                    t.front(), t[t.size() / 4], t[t.size() / 2], t[3 * t.size() / 4], t.back());
        for (float v : times[which]) std::printf(" %.0f", v);
        std::printf("\n");
    }
    return 0;
}
