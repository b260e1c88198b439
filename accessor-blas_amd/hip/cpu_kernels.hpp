// Host-CPU GEMV / dot / TRSV through the HOST side of the accessor, C++14.
//
// The reference has no CPU BLAS kernels (its only CPU numerics are data
// generation and the error metric, SURVEY.md "facts" table), but its README
// states that ranges work the same on CPUs (`/root/reference/README.md:33`).
// These functions are that statement made concrete: the loop bodies are the
// reference kernels' (`cuda/gemv_kernels.cuh:101-110`,
// `cuda/dot_kernels.cuh:153-155`) run serially or over row blocks / chunks
// with std::thread. They exist for `--cpu-baseline` (BASELINE.json config 1
// and the host-CPU baseline of BASELINE.md §3) and are NEVER used in place of
// the GPU path: without the flag the benchmarks only run HIP kernels.
#pragma once

#include <array>
#include <vector>

#include <accessor/range.hpp>
#include <accessor/reduced_row_major.hpp>

#include "benchmark_cli.hpp"
#include "utils.hpp"


template <typename ArType, typename StType>
void cpu_acc_gemv(const matrix_info m_info, ArType alpha, const StType *mtx,
                  const matrix_info x_info, const StType *x,
                  const matrix_info res_info, ArType beta, StType *res,
                  unsigned threads = 1)
{
    constexpr std::size_t dimensionality{2};
    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    using stride_t = std::array<gko::acc::size_type, 1>;
    const auto m_acc = c_range(m_info.size, mtx, stride_t{{m_info.stride}});
    const auto x_acc = c_range(x_info.size, x, stride_t{{x_info.stride}});
    const auto res_acc = range(res_info.size, res, stride_t{{res_info.stride}});
    using ar_type = decltype(alpha * m_acc(0, 0) * x_acc(0, 0) +
                             beta * res_acc(0, 0));
    static_assert(std::is_same<ArType, ar_type>::value, "Types must be equal!");

    parallel_blocks(m_acc.length(0), threads,
                    [&](std::int64_t row_begin, std::int64_t row_end) {
                        const auto num_cols = m_acc.length(1);
                        for (auto row = row_begin; row < row_end; ++row) {
                            ar_type local[4] = {};
                            std::int64_t col = 0;
                            for (; col + 3 < num_cols; col += 4) {
                                for (int u = 0; u < 4; ++u) {
                                    local[u] += m_acc(row, col + u) *
                                                x_acc(col + u, 0);
                                }
                            }
                            for (; col < num_cols; ++col) {
                                local[0] += m_acc(row, col) * x_acc(col, 0);
                            }
                            const ar_type sum =
                                (local[0] + local[1]) + (local[2] + local[3]);
                            if (beta == ArType{0}) {
                                res_acc(row, 0) = alpha * sum;
                            } else {
                                res_acc(row, 0) =
                                    alpha * sum + beta * res_acc(row, 0);
                            }
                        }
                    });
}


template <typename ArType, typename StType, typename ResType>
void cpu_acc_dot(const matrix_info x_info, const StType *x,
                 const matrix_info y_info, const StType *y, ResType *res,
                 unsigned threads = 1)
{
    constexpr std::size_t dimensionality{2};
    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    using stride_t = std::array<gko::acc::size_type, 1>;
    const auto x_acc = c_range(x_info.size, x, stride_t{{x_info.stride}});
    const auto y_acc = c_range(y_info.size, y, stride_t{{y_info.stride}});
    using ar_type = decltype(x_acc(0, 0) + y_acc(0, 0));

    const unsigned parts = threads < 1 ? 1 : threads;
    std::vector<ar_type> partial(parts, ar_type{});
    std::vector<std::thread> pool;
    const std::int64_t n = x_acc.length(0);
    auto body = [&](unsigned t) {
        const std::int64_t begin = n * t / parts;
        const std::int64_t end = n * (t + 1) / parts;
        ar_type local[4] = {};
        std::int64_t i = begin;
        for (; i + 3 < end; i += 4) {
            for (int u = 0; u < 4; ++u) {
                local[u] += x_acc(i + u, 0) * y_acc(i + u, 0);
            }
        }
        for (; i < end; ++i) {
            local[0] += x_acc(i, 0) * y_acc(i, 0);
        }
        partial[t] = (local[0] + local[1]) + (local[2] + local[3]);
    };
    if (parts == 1) {
        body(0);
    } else {
        for (unsigned t = 0; t < parts; ++t) {
            pool.emplace_back(body, t);
        }
        for (auto &worker : pool) {
            worker.join();
        }
    }
    ar_type total{};
    for (unsigned t = 0; t < parts; ++t) {
        total += partial[t];
    }
    *res = static_cast<ResType>(total);
}


/** Plain substitution through the accessor (sequential by nature). */
template <typename ArType, typename StType>
void cpu_acc_trsv(const matrix_info m_info, bool upper, bool unit_diag,
                  const StType *mtx, const matrix_info x_info, StType *x)
{
    constexpr std::size_t dimensionality{2};
    using accessor =
        gko::acc::reduced_row_major<dimensionality, ArType, StType>;
    using range = gko::acc::range<accessor>;
    using c_range = gko::acc::range<typename accessor::const_accessor>;
    using stride_t = std::array<gko::acc::size_type, 1>;
    const auto m_acc = c_range(m_info.size, mtx, stride_t{{m_info.stride}});
    const auto x_acc = range(x_info.size, x, stride_t{{x_info.stride}});
    const std::int64_t n = m_acc.length(0);
    for (std::int64_t k = 0; k < n; ++k) {
        const std::int64_t i = upper ? n - 1 - k : k;
        ArType value = x_acc(i, 0);
        const std::int64_t begin = upper ? i + 1 : 0;
        const std::int64_t end = upper ? n : i;
        for (std::int64_t j = begin; j < end; ++j) {
            value -= m_acc(i, j) * x_acc(j, 0);
        }
        if (!unit_diag) {
            value = value / m_acc(i, i);
        }
        x_acc(i, 0) = value;
    }
}
