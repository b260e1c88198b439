// trsv_benchmark [--error] [--size=N] [additional flags, see --help text]
//
// Drop-in for `/root/reference/cuda/trsv_benchmark.cu:20-236`: same flags, same
// inputs (seed 42, U(-1,1), A then x, in-place LU of the device matrix read
// column-major so that the row-major UPPER triangle holds the bounded unit
// factor), same default system (upper + unit diagonal, `:26-27`), same
// ';'-separated table (ms = min of 10 without a reset of x between runs, or
// with --error the 1-norm relative error against the fp64 plain kernel), same
// column order; vendor columns are rocBLAS instead of cuBLAS.
// Additions: --uplo=upper|lower and --diag=unit|non-unit select the system
// (compile-time constants in the reference); for `lower` the factored matrix
// is transposed once at setup so that the bounded unit factor sits in the
// row-major lower triangle.
#include <cmath>
#include <ios>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "benchmark_cli.hpp"
#include "cpu_kernels.hpp"
#include "memory.hpp"
#include "trsv_kernels.hpp"
#include "trsv_memory.hpp"
#include "utils.hpp"


namespace {


using size_type = matrix_info::size_type;
using column = benchmark_column<matrix_info, matrix_info>;
constexpr char DELIM{';'};

tmtx_t t_matrix_type = tmtx_t::upper;
dmtx_t d_matrix_type = dmtx_t::unit;


TrsvMemory<double> &dataset(TrsvMemory<double> &master,
                            std::unique_ptr<TrsvMemory<double>> &)
{
    return master;
}

template <typename T>
TrsvMemory<T> &dataset(TrsvMemory<double> &master,
                       std::unique_ptr<TrsvMemory<T>> &owned)
{
    owned.reset(new TrsvMemory<T>(master));
    return *owned;
}


struct error_context {
    Memory<double> *cpu_x_ref;
    Memory<double> *reduce_memory;
    double *res_ref_norm;
};

/** `trsv_benchmark.cu:100-115`: fetch x, sum|ref - x| / norm, reset x. */
template <typename T>
std::function<double(matrix_info)> make_error_fn(TrsvMemory<T> &data,
                                                 error_context ctx)
{
    return [&data, ctx](matrix_info x_info) {
        data.sync_x();
        const double error =
            compare(x_info, ctx.cpu_x_ref->const_data(), data.cpu_x_const(),
                    ctx.reduce_memory->data());
        data.reset_x();
        return error / *ctx.res_ref_norm;
    };
}

template <typename T>
void add_plain(std::vector<column> &table, TrsvMemory<T> &data,
               std::uint32_t *helper, std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("TRSV ") + type_name<T>::get(),
        [&data, helper](matrix_info m_info, matrix_info x_info) {
            trsv(m_info, t_matrix_type, d_matrix_type, data.gpu_mtx_const(),
                 x_info, data.gpu_x(), helper);
        },
        err, sizeof(T)});
}

void add_plain(std::vector<column> &, TrsvMemory<float16> &, std::uint32_t *,
               std::function<double(matrix_info)>)
{}

template <typename Ar, typename St>
void add_acc(std::vector<column> &table, TrsvMemory<St> &data,
             std::uint32_t *helper, std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("TRSV Acc<") + type_name<Ar>::get() + ", " +
            type_name<St>::get() + ">",
        [&data, helper](matrix_info m_info, matrix_info x_info) {
            acc_trsv<Ar>(m_info, t_matrix_type, d_matrix_type,
                         data.gpu_mtx_const(), x_info, data.gpu_x(), helper);
        },
        err, sizeof(St)});
}

template <typename St>
void add_acc_same(std::vector<column> &table, TrsvMemory<St> &data,
                  std::uint32_t *helper,
                  std::function<double(matrix_info)> err)
{
    add_acc<St, St>(table, data, helper, err);
}

void add_acc_same(std::vector<column> &, TrsvMemory<float16> &,
                  std::uint32_t *, std::function<double(matrix_info)>)
{}

template <typename T>
void add_vendor(std::vector<column> &table, accblas_vendor_handle *vendor,
                TrsvMemory<T> &data, std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("ROCBLAS TRSV ") + type_name<T>::get(),
        [&data, vendor](matrix_info m_info, matrix_info x_info) {
            rocblas_trsv(vendor, t_matrix_type, d_matrix_type, m_info,
                         data.gpu_mtx_const(), x_info, data.gpu_x());
        },
        err, sizeof(T)});
}

void add_vendor(std::vector<column> &, accblas_vendor_handle *,
                TrsvMemory<float16> &, std::function<double(matrix_info)>)
{}


template <typename ar_type, typename st_type>
int run_gpu(const cli_options &opt)
{
    const size_type max_size = opt.max_size;
    const bool measure_error = opt.error_mode;
    constexpr size_type min_size{100};

    std::default_random_engine rengine(42);
    std::uniform_real_distribution<double> mtx_dist(-1.0, 1.0);
    auto vector_dist = mtx_dist;
    const bool want_lower = t_matrix_type == tmtx_t::lower;
    auto cpu_mtx_gen = [&](matrix_info m_info) {
        return gen_mtx<double>(m_info, mtx_dist, rengine);
    };
    auto cpu_vect_gen = [&](matrix_info v_info) {
        return gen_mtx<double>(v_info, vector_dist, rengine);
    };

    TrsvMemory<double> master(max_size, cpu_mtx_gen, cpu_vect_gen);
    if (want_lower) {
        master.transpose_matrix();
    }
    std::unique_ptr<TrsvMemory<ar_type>> ar_owned;
    std::unique_ptr<TrsvMemory<st_type>> st_owned;
    TrsvMemory<ar_type> &ar_data = dataset(master, ar_owned);
    constexpr bool same_types = std::is_same<ar_type, st_type>::value;
    TrsvMemory<st_type> &st_data =
        same_types ? reinterpret_cast<TrsvMemory<st_type> &>(ar_data)
                   : *(st_owned.reset(new TrsvMemory<st_type>(master)),
                       st_owned.get());

    accblas_vendor_handle *vendor{};
    HIP_CALL(accblas_vendor_create(&vendor));

    const auto max_res_num_elems = master.cpu_x_memory().get_num_elems();
    Memory<double> cpu_x_ref(Memory<double>::Device::cpu, max_res_num_elems);
    Memory<double> reduce_memory(Memory<double>::Device::cpu,
                                 max_res_num_elems);
    double res_ref_norm{1.0};
    Memory<std::uint32_t> trsv_helper(Memory<std::uint32_t>::Device::gpu, 2);
    const error_context ctx{&cpu_x_ref, &reduce_memory, &res_ref_norm};
    const auto master_error = make_error_fn(master, ctx);
    const auto ar_error = make_error_fn(ar_data, ctx);
    const auto st_error = make_error_fn(st_data, ctx);

    // Column order of the reference for <fp64, fp32>; entry 0 (fp64 plain
    // kernel) is the error reference.
    std::vector<column> table;
    constexpr std::size_t benchmark_reference{0};
    std::uint32_t *helper = trsv_helper.data();
    add_plain(table, master, helper, master_error);
    if (!std::is_same<ar_type, double>::value) {
        add_plain(table, ar_data, helper, ar_error);
    }
    if (!same_types) {
        add_plain(table, st_data, helper, st_error);
    }
    add_acc<ar_type, ar_type>(table, ar_data, helper, ar_error);
    if (!same_types) {
        add_acc<ar_type, st_type>(table, st_data, helper, st_error);
        add_acc_same(table, st_data, helper, st_error);
    }
    add_vendor(table, vendor, master, master_error);
    if (!std::is_same<ar_type, double>::value) {
        add_vendor(table, vendor, ar_data, ar_error);
    }
    if (!same_types) {
        add_vendor(table, vendor, st_data, st_error);
    }
    const std::size_t benchmark_num{table.size()};

    std::cout << "Num rows";
    for (const auto &entry : table) {
        std::cout << DELIM << (measure_error ? "Error " : "") << entry.name;
    }
    if (opt.rates && !measure_error) {
        for (const char *what : {"GFLOP/s ", "GB/s ", "%HBM "}) {
            for (const auto &entry : table) {
                std::cout << DELIM << what << entry.name;
            }
        }
    }
    std::cout << '\n';
    std::cout.precision(16);
    std::cout << std::scientific << std::showpos;

    std::vector<double> local_res(benchmark_num);
    const auto start = std::min(max_size, min_size);
    for (const size_type num_rows : sweep_sizes(opt, start, start)) {
        const matrix_info m_info{{{num_rows, num_rows}}, max_size};
        const matrix_info x_info{{{num_rows, 1}}};

        if (measure_error) {
            table[benchmark_reference].run(m_info, x_info);
            master.sync_x();
            cpu_x_ref = master.cpu_x_memory();
            res_ref_norm = reduce<double>(
                x_info, cpu_x_ref.data(),
                [](double a, double b) { return std::abs(a) + std::abs(b); });
            // copy again since the reduction overwrote it
            cpu_x_ref = master.cpu_x_memory();
            master.reset_x();
        }
        for (std::size_t i = 0; i < benchmark_num; ++i) {
            auto local_func = [&]() { table[i].run(m_info, x_info); };
            if (!measure_error) {
                local_res[i] = benchmark_function(local_func, measure_error);
            } else {
                benchmark_function(local_func, measure_error);
                local_res[i] = table[i].error(x_info);
            }
        }

        // a solve whose internal wait gave up leaves 0xdeadbeef in helper[0]
        // (include/accblas.h): its numbers would be meaningless
        std::uint32_t helper_words[2] = {0, 0};
        HIP_CALL(accblas_memcpy_d2h(helper_words, helper, sizeof(helper_words)));
        if (helper_words[0] == 0xdeadbeefu) {
            std::cerr << "trsv_benchmark: a solve at " << num_rows
                      << " rows reported an expired wait (helper[0] = "
                         "0xdeadbeef)\n";
            accblas_vendor_destroy(vendor);
            return 2;
        }
        std::cout << num_rows;
        for (const auto &value : local_res) {
            std::cout << DELIM << value;
        }
        if (opt.rates && !measure_error) {
            // conventions: n^2 flop; s (n(n-1)/2 + 2n)
            // bytes (+ n diagonal elements when they are read)
            const double n = static_cast<double>(num_rows);
            const double elems = n * (n - 1) / 2 + 2 * n +
                                 (d_matrix_type == dmtx_t::non_unit ? n : 0.0);
            for (int what = 0; what < 3; ++what) {
                for (std::size_t i = 0; i < benchmark_num; ++i) {
                    const double seconds = local_res[i] * 1e-3;
                    const double gflops = n * n / seconds / 1e9;
                    const double gbs =
                        static_cast<double>(table[i].storage_bytes) * elems /
                        seconds / 1e9;
                    std::cout << DELIM
                              << (what == 0   ? gflops
                                  : what == 1 ? gbs
                                              : 100.0 * gbs / hbm_peak_gb_per_s);
                }
            }
        }
        std::cout << '\n';
    }
    accblas_vendor_destroy(vendor);
    return 0;
}


}  // namespace


int main(int argc, char **argv)
{
    constexpr size_type default_max_size{24 * 1000};
    constexpr size_type min_size{100};

    // --uplo / --diag are consumed here, the rest goes to the common parser
    std::vector<char *> rest{argv[0]};
    for (int i = 1; i < argc; ++i) {
        const std::string current(argv[i]);
        if (current == "--uplo=lower") {
            t_matrix_type = tmtx_t::lower;
        } else if (current == "--uplo=upper") {
            t_matrix_type = tmtx_t::upper;
        } else if (current == "--diag=unit") {
            d_matrix_type = dmtx_t::unit;
        } else if (current == "--diag=non-unit") {
            d_matrix_type = dmtx_t::non_unit;
        } else {
            rest.push_back(argv[i]);
        }
    }
    cli_options opt;
    const cli_texts texts{
        "compute errors of TRSV",
        "set the maximum size (used for both rows and cols) of the matrix.",
        "benchmark different TRSVs", "The matrix size needs to be at least "};
    const int exit_code = parse_cli(static_cast<int>(rest.size()), rest.data(),
                                    opt, default_max_size, min_size, texts);
    if (exit_code >= 0) {
        return exit_code;
    }
    if (opt.cpu_baseline) {
        std::cerr << "--cpu-baseline is not available for TRSV: its inputs "
                     "need the GPU LU factorisation\n";
        return 1;
    }
    return dispatch_types(opt, [&](auto ar, auto st) {
        return run_gpu<decltype(ar), decltype(st)>(opt);
    });
}
