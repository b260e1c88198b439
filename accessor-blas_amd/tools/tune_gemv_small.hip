// Developer tool (not part of the library): small GEMVs (the first third of the
// reference's sweep, cuda/gemv_benchmark.cu:211-214: n = 100, 200, ... inside a
// matrix of row stride 24500) for a list of gemv_vec shapes, with the rocBLAS
// column of the CLI timed in the same process. Protocol = benchmark_function
// (cuda/utils.cuh:236-262): warm-up, then the minimum of 10 event-timed single
// launches. Usage: tune_gemv_small [lda] [reps]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "gemv_kernels.hpp"
#include "../../include/accblas.h"

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
    std::function<void(long, long)> run;  // arguments: rows, cols
};

static hipEvent_t ev_start, ev_stop;

static bool fresh_events = false;  // as the CLI: new events for every run

double time_us(const std::function<void()> &f, int reps)
{
    f();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        if (fresh_events) {
            // `CudaTimer` of the reference destroys and re-creates both events
            // after every measurement (cuda/utils.cuh:144-182)
            CHECK(hipEventDestroy(ev_start));
            CHECK(hipEventDestroy(ev_stop));
            CHECK(hipEventCreate(&ev_start));
            CHECK(hipEventCreate(&ev_stop));
        }
        CHECK(hipEventRecord(ev_start, nullptr));
        f();
        CHECK(hipEventRecord(ev_stop, nullptr));
        CHECK(hipEventSynchronize(ev_stop));
        float ms;
        CHECK(hipEventElapsedTime(&ms, ev_start, ev_stop));
        best = std::min(best, double(ms) * 1e3);
    }
    return best;
}

static bool fine_pass = false;
// the reference allocates max_size x max_size (2.4 GB of fp32 at 24500): small
// blocks of a large allocation run slower than the same blocks of a small one
static bool big_matrix = false;
// "mall": production and two finer shapes with and without the non-temporal
// hint on the matrix loads, n = 3000 ... 12000 (the matrix fits the 256 MiB
// Infinity Cache up to n = 8000 with 4-byte and 5600 with 8-byte storage)
static bool mall_pass = false;

template <typename Ar, typename St>
void sweep(const char *title, long lda, int reps, rocblas_handle blas)
{
    using accessor = gko::acc::reduced_row_major<2, Ar, St>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    using sz = std::array<gko::acc::size_type, 2>;
    using st = std::array<gko::acc::size_type, 1>;
    const long max_n = big_matrix ? lda : (fine_pass ? 24000 : (mall_pass ? 12000 : 6000));
    St *A, *x, *res;
    CHECK(hipMalloc(&A, size_t(max_n) * lda * sizeof(St)));
    CHECK(hipMalloc(&x, lda * sizeof(St)));
    CHECK(hipMalloc(&res, lda * sizeof(St)));
    fill<<<4096, 256>>>(A, size_t(max_n) * lda, 1);
    fill<<<64, 256>>>(x, size_t(lda), 2);
    CHECK(hipMemset(res, 0, lda * sizeof(St)));

    std::vector<entry> entries;
#define ADD(W, WPR, R, U)                                                      \
    entries.push_back({"w" #W "p" #WPR "r" #R "u" #U, [=](long n, long nc) {   \
                           c_range m(sz{{n, nc}}, (const St *)A, st{{lda}});   \
                           c_range xv(sz{{nc, 1}}, (const St *)x, st{{1}});    \
                           range rv(sz{{n, 1}}, res, st{{1}});                 \
                           CHECK((launch_gemv<gemv_config<W, WPR, R, U, true>>( \
                               Ar(1), m, xv, Ar(1), rv, nullptr,               \
                               gemv_skew_mode::never)));                       \
                       }});
#define ADDC(W, WPR, R, U)                                                     \
    entries.push_back({"w" #W "p" #WPR "r" #R "u" #U "c", [=](long n, long nc) { \
                           c_range m(sz{{n, nc}}, (const St *)A, st{{lda}});   \
                           c_range xv(sz{{nc, 1}}, (const St *)x, st{{1}});    \
                           range rv(sz{{n, 1}}, res, st{{1}});                 \
                           CHECK((launch_gemv<gemv_config<W, WPR, R, U, false>>( \
                               Ar(1), m, xv, Ar(1), rv, nullptr,               \
                               gemv_skew_mode::never)));                       \
                       }});
    ADD(8, 4, 4, 2)  // production shape
    if (mall_pass) {
        ADDC(8, 4, 4, 2)
        ADD(8, 4, 2, 2)
        ADDC(8, 4, 2, 2)
        ADD(4, 2, 2, 2)
        ADDC(4, 2, 2, 2)
        ADD(4, 4, 2, 1)
        ADDC(4, 4, 2, 1)
        ADD(2, 2, 1, 2)
        ADDC(2, 2, 1, 2)
    } else if (big_matrix) {
        ADD(2, 2, 1, 2)
        ADD(2, 2, 1, 4)
        ADD(2, 2, 1, 8)
        ADD(1, 1, 1, 4)
        ADD(1, 1, 1, 8)
        ADD(4, 4, 1, 2)
        ADD(4, 4, 1, 4)
        ADD(4, 2, 2, 2)
        ADD(4, 2, 2, 4)
        ADD(2, 1, 1, 8)
        ADD(4, 2, 1, 4)
    } else if (fine_pass) {
        ADD(2, 2, 1, 2)
        ADD(1, 1, 1, 4)
        ADD(4, 4, 1, 1)
        ADD(4, 2, 1, 2)
        ADD(4, 2, 2, 1)
        ADD(4, 2, 2, 2)
        ADD(4, 4, 2, 1)
        ADD(8, 2, 1, 2)
        ADD(8, 4, 2, 2)
        ADD(8, 4, 1, 2)
        ADD(8, 2, 2, 2)
    } else {
    ADD(8, 4, 4, 1)
    ADD(8, 2, 2, 2)
    ADD(8, 2, 2, 1)
    ADD(8, 2, 1, 2)
    ADD(8, 1, 1, 2)
    ADD(8, 1, 1, 4)
    ADD(4, 1, 1, 2)
    ADD(4, 1, 1, 4)
    ADD(4, 1, 2, 2)
    ADD(4, 2, 1, 2)
    ADD(4, 2, 2, 1)
    ADD(4, 2, 2, 2)
    ADD(4, 4, 1, 1)
    ADD(4, 4, 1, 2)
    ADD(4, 4, 2, 1)
    ADD(2, 1, 1, 4)
    ADD(2, 2, 1, 2)
    ADD(2, 2, 1, 4)
    ADD(1, 1, 1, 4)
    ADD(1, 1, 2, 2)
    }
#undef ADD
#undef ADDC
    // the same call through the C ABI of the library (what the CLI times)
    entries.push_back({"library", [=](long n, long nc) {
        const accblas_dtype ar = sizeof(Ar) == 8 ? ACCBLAS_F64 : ACCBLAS_F32;
        const accblas_dtype st_t = sizeof(St) == 8 ? ACCBLAS_F64 : ACCBLAS_F32;
        if (accblas_acc_gemv(ar, st_t, n, nc, 1.0, A, lda, x, 1, 1.0, res, 1, nullptr) != 0) {
            std::fprintf(stderr, "accblas_acc_gemv: %s\n", accblas_last_error_string());
            std::exit(1);
        }
    }});
    entries.push_back({"rocblas", [=](long n, long nc) {
        if (sizeof(St) == 4 && sizeof(Ar) == 4) {
            const float one = 1.0f;
            rocblas_sgemv(blas, rocblas_operation_transpose, int(nc), int(n), &one,
                          (const float *)A, int(lda), (const float *)x, 1, &one,
                          (float *)res, 1);
        } else if (sizeof(St) == 8) {
            const double one = 1.0;
            rocblas_dgemv(blas, rocblas_operation_transpose, int(nc), int(n), &one,
                          (const double *)A, int(lda), (const double *)x, 1, &one,
                          (double *)res, 1);
        } else {
            // mixed precision: the comparison column is rocBLAS in the STORAGE type
            const float one = 1.0f;
            rocblas_sgemv(blas, rocblas_operation_transpose, int(nc), int(n), &one,
                          (const float *)A, int(lda), (const float *)x, 1, &one,
                          (float *)res, 1);
        }
    }});

    const std::vector<long> sizes =
        mall_pass ? std::vector<long>{1000, 2000, 3000, 3500, 4000, 4500, 5000, 5500, 6000, 6500, 7000,
                                      7500, 8000, 8500, 9000, 10000, 11000, 12000}
        : fine_pass ? std::vector<long>{2000, 3000, 3100, 3300, 3500, 4000, 4100, 4500, 5000, 5200,
                                      5500, 6000, 6200, 6500, 7000, 8000, 8200, 10000, -1000,
                                      -2000, -3000, -4000}
                  : std::vector<long>{100,  500,  900,  1100, 1300, 1600, 1900, 2200,
                                      2500, 3000, 3500, 4000, 4100, 4500, 5000, 6000};
    // (a negative size -c stands for the tall case 24000 rows x c columns)
    std::printf("== %s, lda = %ld, min of %d single launches, us ==\n%-10s", title, lda,
                reps, "shape");
    for (long n : sizes) {
        std::printf(n < 0 ? " T%6ld" : "%7ld", n < 0 ? -n : n);
    }
    std::printf("\n");
    // two passes, alternating the order, keep the minimum
    std::vector<std::vector<double>> best(entries.size(),
                                          std::vector<double>(sizes.size(), 1e30));
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t s = 0; s < sizes.size(); ++s) {
            for (size_t i = 0; i < entries.size(); ++i) {
                const size_t k = pass == 0 ? i : entries.size() - 1 - i;
                const long n = sizes[s] < 0 ? 24000 : sizes[s];
                const long nc = sizes[s] < 0 ? -sizes[s] : sizes[s];
                best[k][s] = std::min(best[k][s],
                                      time_us([&]() { entries[k].run(n, nc); }, reps));
            }
        }
    }
    for (size_t i = 0; i < entries.size(); ++i) {
        std::printf("%-10s", entries[i].name.c_str());
        for (size_t s = 0; s < sizes.size(); ++s) {
            std::printf("%7.2f", best[i][s]);
        }
        std::printf("\n");
    }
    CHECK(hipFree(A));
    CHECK(hipFree(x));
    CHECK(hipFree(res));
}

int main(int argc, char **argv)
{
    const long lda = argc > 1 ? std::atol(argv[1]) : 24500;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 10;
    fine_pass = argc > 3 && std::string(argv[3]) == "fine";
    fresh_events = argc > 3 && std::string(argv[3]) == "fresh";
    big_matrix = argc > 3 && std::string(argv[3]) == "big";
    mall_pass = argc > 3 && std::string(argv[3]) == "mall";
    CHECK(hipEventCreate(&ev_start));
    CHECK(hipEventCreate(&ev_stop));
    rocblas_handle blas;
    rocblas_create_handle(&blas);
    sweep<double, float>("Acc<fp64, fp32> (rocblas = sgemv)", lda, reps, blas);
    sweep<double, double>("fp64 (rocblas = dgemv)", lda, reps, blas);
    sweep<float, float>("fp32 (rocblas = sgemv)", lda, reps, blas);
    rocblas_destroy_handle(blas);
    return 0;
}
