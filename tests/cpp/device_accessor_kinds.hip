// GPU test (compiled and run by tests/test_gpu_accessor_kinds.py): the kernels
// of accessor-blas_amd/csrc work with the other accessor kinds of
// accessor-blas_amd/include/accessor ("all others work accordingly",
// /root/reference/README.md:19):
//   - range<row_major<double, 2>>: plain storage, takes the 16-byte vector paths
//     of GEMV and dot and the TRSV kernel;
//   - range<scaled_reduced_row_major<2, double, int32_t, 0b10>>: value =
//     storage * scalar[column]; only reachable through operator(), so GEMV and
//     dot run their generic paths.
// Expected values come from host loops over the SAME accessor types on host
// copies of the data (the headers are host + device).
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>
#include <accessor/row_major.hpp>
#include <accessor/scaled_reduced_row_major.hpp>

#include "dot_kernels.hpp"
#include "gemv_kernels.hpp"
#include "trsv_kernels.hpp"

#define CHECK(call)                                                      \
    do {                                                                 \
        const hipError_t err_ = (call);                                  \
        if (err_ != hipSuccess) {                                        \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,       \
                         hipGetErrorString(err_));                       \
            std::exit(2);                                                \
        }                                                                \
    } while (false)

namespace {

using gko::acc::size_type;
using size2 = std::array<size_type, 2>;
using stride1 = std::array<size_type, 1>;
using namespace accblas::dev;

template <typename T>
T *to_device(const std::vector<T> &host)
{
    T *ptr{};
    CHECK(hipMalloc(&ptr, host.size() * sizeof(T)));
    CHECK(hipMemcpy(ptr, host.data(), host.size() * sizeof(T),
                    hipMemcpyHostToDevice));
    return ptr;
}

template <typename T>
std::vector<T> to_host(const T *ptr, std::size_t count)
{
    std::vector<T> host(count);
    CHECK(hipMemcpy(host.data(), ptr, count * sizeof(T), hipMemcpyDeviceToHost));
    return host;
}

double rel_error(const std::vector<double> &want, const std::vector<double> &got)
{
    double num = 0, den = 0;
    for (std::size_t i = 0; i < want.size(); ++i) {
        num += std::abs(want[i] - got[i]);
        den += std::abs(want[i]);
    }
    return den > 0 ? num / den : num;
}

void expect(bool ok, const char *what, double value)
{
    std::printf("%-58s %.3e %s\n", what, value, ok ? "ok" : "FAILED");
    if (!ok) {
        std::exit(1);
    }
}

dot_workspace make_dot_workspace()
{
    dot_workspace ws{};
    CHECK(hipMalloc(&ws.slots, dot_slot_bytes));
    CHECK(hipMemset(ws.slots, 0, dot_slot_bytes));
    CHECK(hipMalloc(&ws.scalar, 16));
    ws.serial = 1;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    ws.num_cus = prop.multiProcessorCount;
    return ws;
}


void row_major_cases(std::default_random_engine &engine)
{
    using accessor = gko::acc::row_major<double, 2>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<accessor::const_accessor>;
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    const size_type rows = 301, cols = 1030, lda = 1032;
    std::vector<double> a(rows * lda), x(cols), res(rows);
    for (auto &v : a) v = dist(engine);
    for (auto &v : x) v = dist(engine);
    for (auto &v : res) v = dist(engine);
    // host expectation through the same accessor
    std::vector<double> want = res;
    {
        c_range m(size2{{rows, cols}}, a.data(), stride1{{lda}});
        c_range xv(size2{{cols, 1}}, x.data(), stride1{{1}});
        range r(size2{{rows, 1}}, want.data(), stride1{{1}});
        for (size_type i = 0; i < rows; ++i) {
            long double sum = 0;
            for (size_type j = 0; j < cols; ++j) {
                sum += static_cast<long double>(m(i, j)) * xv(j, 0);
            }
            r(i, 0) = static_cast<double>(
                0.5L * sum - 2.0L * static_cast<double>(r(i, 0)));
        }
    }
    double *d_a = to_device(a), *d_x = to_device(x), *d_res = to_device(res);
    {
        c_range m(size2{{rows, cols}}, d_a, stride1{{lda}});
        c_range xv(size2{{cols, 1}}, d_x, stride1{{1}});
        range r(size2{{rows, 1}}, d_res, stride1{{1}});
        static_assert(view_traits<c_range>::is_pointer_view, "row_major is plain");
        CHECK(launch_gemv(0.5, m, xv, -2.0, r, nullptr));
        CHECK(hipDeviceSynchronize());
    }
    expect(rel_error(want, to_host(d_res, rows)) <= 1e-14,
           "GEMV on range<row_major<double, 2>> (vector path)",
           rel_error(want, to_host(d_res, rows)));

    // dot of two row_major vectors
    {
        const size_type n = 100003;
        std::vector<double> u(n), v(n);
        long double sum = 0;
        for (size_type i = 0; i < n; ++i) {
            u[i] = dist(engine);
            v[i] = dist(engine);
            sum += static_cast<long double>(u[i]) * v[i];
        }
        double *d_u = to_device(u), *d_v = to_device(v);
        double *d_out{};
        CHECK(hipMalloc(&d_out, sizeof(double)));
        auto ws = make_dot_workspace();
        c_range uv(size2{{n, 1}}, d_u, stride1{{1}});
        c_range vv(size2{{n, 1}}, d_v, stride1{{1}});
        CHECK((launch_dot<double>(ws, false, uv, vv, d_out, nullptr)));
        CHECK(hipDeviceSynchronize());
        const double got = to_host(d_out, 1)[0];
        const double err = std::abs(got - static_cast<double>(sum));
        expect(err <= 64 * 2.2e-16 * std::sqrt(static_cast<double>(n)) / 3,
               "dot on range<row_major<double, 2>>", err);
    }

    // unit lower TRSV on a row_major matrix, manufactured solution
    {
        const size_type n = 777, ld = 800;
        std::vector<double> t(n * ld), x_true(n), b(n);
        for (auto &v : t) v = dist(engine) / 16.0;
        for (auto &v : x_true) v = dist(engine);
        for (size_type i = 0; i < n; ++i) {
            long double sum = x_true[i];  // unit diagonal
            for (size_type j = 0; j < i; ++j) {
                sum += static_cast<long double>(t[i * ld + j]) * x_true[j];
            }
            b[i] = static_cast<double>(sum);
        }
        double *d_t = to_device(t), *d_b = to_device(b);
        std::uint32_t *helper{};
        CHECK(hipMalloc(&helper, 8));
        const std::size_t ws_bytes = trsv_workspace_bytes<double>(n);
        granule_word *granules{};
        CHECK(hipMalloc(&granules, ws_bytes));
        CHECK(hipMemset(granules, 0, ws_bytes));
        trsv_state *state{};
        CHECK(hipMalloc(&state, 64));
        const trsv_state initial{trsv_clean_tickets, 0u, trsv_first_epoch, 0u, 0u};
        CHECK(hipMemcpy(state, &initial, sizeof(initial), hipMemcpyHostToDevice));
        const trsv_session session{
            state, granules, static_cast<index_type>(ws_bytes / 8),
            trsv_next_serial()};
        c_range m(size2{{n, n}}, d_t, stride1{{ld}});
        range xv(size2{{n, 1}}, d_b, stride1{{1}});
        CHECK((launch_trsv<double>(false, true, m, xv, helper, session, nullptr)));
        CHECK(hipDeviceSynchronize());
        expect(rel_error(x_true, to_host(d_b, n)) <= 1e-12,
               "TRSV on range<row_major<double, 2>>",
               rel_error(x_true, to_host(d_b, n)));
        expect(to_host(helper, 1)[0] == 0xffffffffu, "TRSV helper word", 0.0);
    }
}


void scaled_cases(std::default_random_engine &engine)
{
    // A(i, j) = double(int32 storage) * scalar[j]: one scalar per column
    using mtx_accessor =
        gko::acc::scaled_reduced_row_major<2, double, std::int32_t, 0b10>;
    using c_mtx = gko::acc::range<mtx_accessor::const_accessor>;
    using vec_accessor = gko::acc::reduced_row_major<2, double, float>;
    using vec = gko::acc::range<vec_accessor>;
    using c_vec = gko::acc::range<vec_accessor::const_accessor>;
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::uniform_int_distribution<std::int32_t> ints(-(1 << 20), 1 << 20);
    const size_type rows = 257, cols = 700, lda = 701;
    std::vector<std::int32_t> a(rows * lda);
    std::vector<double> scale(cols);
    std::vector<float> x(cols), res(rows);
    for (auto &v : a) v = ints(engine);
    for (auto &v : scale) v = (1.0 + dist(engine) * 0.5) / (1 << 20);
    for (auto &v : x) v = static_cast<float>(dist(engine));
    for (auto &v : res) v = static_cast<float>(dist(engine));
    std::vector<float> want = res;
    {
        c_mtx m(size2{{rows, cols}}, a.data(), stride1{{lda}}, scale.data());
        c_vec xv(size2{{cols, 1}}, x.data(), stride1{{1}});
        vec r(size2{{rows, 1}}, want.data(), stride1{{1}});
        for (size_type i = 0; i < rows; ++i) {
            long double sum = 0;
            for (size_type j = 0; j < cols; ++j) {
                sum += static_cast<long double>(m(i, j)) * xv(j, 0);
            }
            r(i, 0) = static_cast<double>(
                1.5L * sum + 0.25L * static_cast<double>(r(i, 0)));
        }
    }
    auto *d_a = to_device(a);
    auto *d_scale = to_device(scale);
    auto *d_x = to_device(x);
    auto *d_res = to_device(res);
    {
        c_mtx m(size2{{rows, cols}}, d_a, stride1{{lda}}, d_scale);
        c_vec xv(size2{{cols, 1}}, d_x, stride1{{1}});
        vec r(size2{{rows, 1}}, d_res, stride1{{1}});
        static_assert(!view_traits<c_mtx>::is_pointer_view,
                      "scaled values are not plain storage");
        CHECK(launch_gemv(1.5, m, xv, 0.25, r, nullptr));
        CHECK(hipDeviceSynchronize());
    }
    const auto got = to_host(d_res, rows);
    const double err = rel_error(std::vector<double>(want.begin(), want.end()),
                                 std::vector<double>(got.begin(), got.end()));
    expect(err <= 1e-7, "GEMV on range<scaled_reduced_row_major<2, f64, i32, 0b10>>",
           err);

    // dot of a scaled int16 vector (one scalar for the whole column) with a
    // reduced fp32 vector
    {
        using sv_accessor =
            gko::acc::scaled_reduced_row_major<2, double, std::int16_t, 0b10>;
        using c_sv = gko::acc::range<sv_accessor::const_accessor>;
        const size_type n = 65537;
        std::uniform_int_distribution<int> shorts(-32000, 32000);
        std::vector<std::int16_t> u(n);
        std::vector<float> v(n);
        const std::vector<double> one_scale{1.0 / 32000.0};
        long double sum = 0, scale_sq = 0;
        for (size_type i = 0; i < n; ++i) {
            u[i] = static_cast<std::int16_t>(shorts(engine));
            v[i] = static_cast<float>(dist(engine));
            const long double term =
                static_cast<long double>(u[i]) * one_scale[0] * v[i];
            sum += term;
            scale_sq += term * term;
        }
        auto *d_u = to_device(u);
        auto *d_v = to_device(v);
        auto *d_s = to_device(one_scale);
        double *d_out{};
        CHECK(hipMalloc(&d_out, sizeof(double)));
        auto ws = make_dot_workspace();
        c_sv uv(size2{{n, 1}}, d_u, stride1{{1}}, d_s);
        c_vec vv(size2{{n, 1}}, d_v, stride1{{1}});
        CHECK((launch_dot<double>(ws, false, uv, vv, d_out, nullptr)));
        CHECK(hipDeviceSynchronize());
        const double gpu = to_host(d_out, 1)[0];
        const double abs_err = std::abs(gpu - static_cast<double>(sum));
        expect(abs_err <= 64 * 2.2e-16 * std::sqrt(static_cast<double>(scale_sq)),
               "dot on scaled_reduced_row_major<.., i16, 0b10> x reduced fp32",
               abs_err);
    }
}

}  // namespace


int main()
{
    std::default_random_engine engine(42);
    row_major_cases(engine);
    scaled_cases(engine);
    std::printf("device accessor kinds ok\n");
    return 0;
}
