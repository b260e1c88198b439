// gemv_benchmark [--error] [--size=N] [additional flags, see --help text]
//
// Drop-in for `/root/reference/cuda/gemv_benchmark.cu:19-251`: same flags, same
// inputs (seed 42, U(-1,1), A then x then res, top-left n x n block of the
// max_size matrix, alpha = beta = 1), same ';'-separated table on stdout
// (time in ms = min of 10, or with --error the 1-norm relative error against
// the fp64 plain kernel), same column order; the two vendor columns are
// rocBLAS instead of cuBLAS. Host code is C++14 and reaches the gfx950 kernels
// only through the C ABI.
#include <cmath>
#include <ios>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "benchmark_cli.hpp"
#include "cpu_kernels.hpp"
#include "gemv_kernels.hpp"
#include "gemv_memory.hpp"
#include "memory.hpp"
#include "utils.hpp"


namespace {


using size_type = matrix_info::size_type;
using column = benchmark_column<matrix_info, matrix_info, matrix_info>;
constexpr char DELIM{';'};
constexpr double master_alpha{1.0};
constexpr double master_beta{1.0};


// the fp64 master doubles as the arithmetic-type data set when ar == fp64
GemvMemory<double> &dataset(GemvMemory<double> &master,
                            std::unique_ptr<GemvMemory<double>> &)
{
    return master;
}

template <typename T>
GemvMemory<T> &dataset(GemvMemory<double> &master,
                       std::unique_ptr<GemvMemory<T>> &owned)
{
    owned.reset(new GemvMemory<T>(master));
    return *owned;
}


struct error_context {
    bool measure_error;
    bool reset_result;
    Memory<double> *cpu_res_ref;
    Memory<double> *reduce_memory;
    double *res_ref_norm;
};


/** Error of `data`'s device result against the reference result, then
 *  restores the initial result (`gemv_benchmark.cu:101-124`). */
template <typename T>
std::function<double(matrix_info)> make_error_fn(
    GemvMemory<T> &data, std::shared_ptr<Memory<T>> init, error_context ctx)
{
    return [&data, init, ctx](matrix_info x_info) {
        double error{};
        if (ctx.measure_error) {
            data.sync_result();
            error = compare(x_info, ctx.cpu_res_ref->const_data(),
                            data.cpu_res_const(), ctx.reduce_memory->data());
        }
        if (ctx.reset_result) {
            data.gpu_res_memory() = *init;
        }
        return error / *ctx.res_ref_norm;
    };
}


template <typename T>
void add_plain(std::vector<column> &table, GemvMemory<T> &data,
               std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("GEMV ") + type_name<T>::get(),
        [&data](matrix_info m, matrix_info x, matrix_info r) {
            gemv(m, static_cast<T>(master_alpha), data.gpu_mtx_const(), x,
                 data.gpu_x_const(), r, static_cast<T>(master_beta),
                 data.gpu_res());
        },
        err, sizeof(T)});
}

// there is no fp16 ARITHMETIC: no plain fp16 kernel, no Acc<fp16, fp16>
void add_plain(std::vector<column> &, GemvMemory<float16> &,
               std::function<double(matrix_info)>)
{}

template <typename Ar, typename St>
void add_acc(std::vector<column> &table, GemvMemory<St> &data,
             std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("GEMV Acc<") + type_name<Ar>::get() + ", " +
            type_name<St>::get() + ">",
        [&data](matrix_info m, matrix_info x, matrix_info r) {
            acc_gemv<Ar>(m, static_cast<Ar>(master_alpha),
                         data.gpu_mtx_const(), x, data.gpu_x_const(), r,
                         static_cast<Ar>(master_beta), data.gpu_res());
        },
        err, sizeof(St)});
}

template <typename St>
void add_acc_same(std::vector<column> &table, GemvMemory<St> &data,
                  std::function<double(matrix_info)> err)
{
    add_acc<St, St>(table, data, err);
}

void add_acc_same(std::vector<column> &, GemvMemory<float16> &,
                  std::function<double(matrix_info)>)
{}

template <typename T>
void add_vendor(std::vector<column> &table, accblas_vendor_handle *vendor,
                GemvMemory<T> &data, std::function<double(matrix_info)> err)
{
    table.push_back(column{
        std::string("ROCBLAS GEMV ") + type_name<T>::get(),
        [&data, vendor](matrix_info m, matrix_info x, matrix_info r) {
            rocblas_gemv(vendor, m, static_cast<T>(master_alpha),
                         data.gpu_mtx_const(), x, data.gpu_x_const(), r,
                         static_cast<T>(master_beta), data.gpu_res());
        },
        err, sizeof(T)});
}

void add_vendor(std::vector<column> &, accblas_vendor_handle *,
                GemvMemory<float16> &, std::function<double(matrix_info)>)
{}


template <typename ar_type, typename st_type>
int run_gpu(const cli_options &opt)
{
    const size_type max_size = opt.max_size;
    const bool measure_error = opt.error_mode;
    // The result vector only needs a reset when errors are measured and
    // beta != 0 (otherwise its initial content does not matter).
    const bool reset_result{measure_error && master_beta != 0};

    std::default_random_engine rengine(42);
    std::uniform_real_distribution<double> mtx_dist(-1.0, 1.0);
    auto vector_dist = mtx_dist;

    // fp64 master data; the other precisions are static_casts of it
    GemvMemory<double> master(max_size, mtx_dist, vector_dist, rengine);
    std::unique_ptr<GemvMemory<ar_type>> ar_owned;
    std::unique_ptr<GemvMemory<st_type>> st_owned;
    GemvMemory<ar_type> &ar_data = dataset(master, ar_owned);
    constexpr bool same_types = std::is_same<ar_type, st_type>::value;
    GemvMemory<st_type> &st_data =
        same_types ? reinterpret_cast<GemvMemory<st_type> &>(ar_data)
                   : *(st_owned.reset(new GemvMemory<st_type>(master)),
                       st_owned.get());

    accblas_vendor_handle *vendor{};
    HIP_CALL(accblas_vendor_create(&vendor));

    const auto max_res_num_elems = master.cpu_res_memory().get_num_elems();
    Memory<double> cpu_res_ref(Memory<double>::Device::cpu, max_res_num_elems);
    Memory<double> reduce_memory(Memory<double>::Device::cpu,
                                 max_res_num_elems);
    double res_ref_norm{1.0};
    const error_context ctx{measure_error, reset_result, &cpu_res_ref,
                            &reduce_memory, &res_ref_norm};
    auto master_init =
        std::make_shared<Memory<double>>(master.cpu_res_memory());
    auto ar_init =
        std::make_shared<Memory<ar_type>>(ar_data.cpu_res_memory());
    auto st_init =
        std::make_shared<Memory<st_type>>(st_data.cpu_res_memory());
    const auto master_error = make_error_fn(master, master_init, ctx);
    const auto ar_error = make_error_fn(ar_data, ar_init, ctx);
    const auto st_error = make_error_fn(st_data, st_init, ctx);

    // Column order of the reference for <fp64, fp32>: plain ar, plain st,
    // Acc<ar, ar>, Acc<ar, st>, Acc<st, st>, vendor ar, vendor st. Entry 0 is
    // the error reference and is always the fp64 plain kernel.
    std::vector<column> table;
    constexpr size_type benchmark_reference{0};
    add_plain(table, master, master_error);
    if (!std::is_same<ar_type, double>::value) {
        add_plain(table, ar_data, ar_error);
    }
    if (!same_types) {
        add_plain(table, st_data, st_error);
    }
    add_acc<ar_type, ar_type>(table, ar_data, ar_error);
    if (!same_types) {
        add_acc<ar_type, st_type>(table, st_data, st_error);
        add_acc_same(table, st_data, st_error);
    }
    add_vendor(table, vendor, master, master_error);
    if (!std::is_same<ar_type, double>::value) {
        add_vendor(table, vendor, ar_data, ar_error);
    }
    if (!same_types) {
        add_vendor(table, vendor, st_data, st_error);
    }
    const size_type benchmark_num = static_cast<size_type>(table.size());

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
    // showpos: explicit + sign for positive numbers
    std::cout << std::scientific << std::showpos;

    std::vector<double> local_res(benchmark_num);
    constexpr size_type default_start{100};
    for (const size_type num_rows : sweep_sizes(opt, default_start, default_start)) {
        const matrix_info m_info{{{num_rows, num_rows}}, max_size};
        const matrix_info x_info{{{num_rows, 1}}};
        const matrix_info res_info{{{num_rows, 1}}};

        if (measure_error) {
            table[benchmark_reference].run(m_info, x_info, res_info);
            cpu_res_ref.copy_from(master.gpu_res_memory());
            res_ref_norm = reduce<double>(
                res_info, cpu_res_ref.data(),
                [](double a, double b) { return std::abs(a) + std::abs(b); });
            // copy again since the reduction overwrote it
            cpu_res_ref.copy_from(master.gpu_res_memory());
            master.gpu_res_memory().copy_from(*master_init);
        }
        for (size_type i = 0; i < benchmark_num; ++i) {
            auto local_func = [&]() { table[i].run(m_info, x_info, res_info); };
            if (!measure_error) {
                local_res[i] = benchmark_function(local_func, measure_error);
            } else {
                benchmark_function(local_func, measure_error);
                local_res[i] = table[i].error(x_info);
            }
        }

        std::cout << num_rows;
        for (const auto &value : local_res) {
            std::cout << DELIM << value;
        }
        if (opt.rates && !measure_error) {
            // conventions: 2 n^2 flop; s (n^2 + 3 n) bytes
            const double n = static_cast<double>(num_rows);
            for (int what = 0; what < 3; ++what) {
                for (size_type i = 0; i < benchmark_num; ++i) {
                    const double seconds = local_res[i] * 1e-3;
                    const double gflops = 2.0 * n * n / seconds / 1e9;
                    const double gbs = static_cast<double>(table[i].storage_bytes) *
                                       (n * n + 3.0 * n) / seconds / 1e9;
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


/** Host-only path: the same sweep with the CPU accessor kernels, 1 thread and
 *  all hardware threads, plus the reference's CPU error check timed on the
 *  result vector. Touches no GPU. */
template <typename ar_type, typename st_type>
int run_cpu_baseline(const cli_options &opt)
{
    const size_type max_size = opt.max_size;
    std::default_random_engine rengine(42);
    std::uniform_real_distribution<double> mtx_dist(-1.0, 1.0);
    auto vector_dist = mtx_dist;
    const matrix_info full_m{{{max_size, max_size}}};
    const matrix_info full_v{{{max_size, 1}}};
    auto mtx = gen_mtx<double>(full_m, mtx_dist, rengine);
    auto x = gen_mtx<double>(full_v, vector_dist, rengine);
    auto res = gen_mtx<double>(full_v, vector_dist, rengine);
    Memory<st_type> st_mtx(Memory<st_type>::Device::cpu, full_m.get_1d_size());
    Memory<st_type> st_x(Memory<st_type>::Device::cpu, max_size);
    Memory<st_type> st_res(Memory<st_type>::Device::cpu, max_size);
    const auto cast = [](double v) { return static_cast<st_type>(v); };
    convert_mtx(full_m, mtx.const_data(), st_mtx.data(), cast);
    convert_mtx(full_v, x.const_data(), st_x.data(), cast);
    convert_mtx(full_v, res.const_data(), st_res.data(), cast);
    Memory<double> work64(res);
    Memory<st_type> work_st(st_res);
    Memory<double> tmp(Memory<double>::Device::cpu, max_size);
    const unsigned cores = std::max(1u, std::thread::hardware_concurrency());

    const std::string acc_name = std::string("Acc<") +
                                 type_name<ar_type>::get() + ", " +
                                 type_name<st_type>::get() + ">";
    std::cout << "Num rows" << DELIM << "CPU GEMV fp64 1 core" << DELIM
              << "CPU GEMV fp64 " << cores << " cores" << DELIM << "CPU GEMV "
              << acc_name << " 1 core" << DELIM << "CPU GEMV " << acc_name
              << ' ' << cores << " cores" << DELIM
              << "CPU error check (compare+reduce) 1 core" << DELIM << "Error "
              << acc_name << '\n';
    std::cout.precision(16);
    std::cout << std::scientific << std::showpos;
    for (const size_type n : sweep_sizes(opt, 100, 100)) {
        const matrix_info m_info{{{n, n}}, max_size};
        const matrix_info v_info{{{n, 1}}};
        auto run64 = [&](unsigned threads) {
            work64.copy_from(res);
            cpu_acc_gemv<double>(m_info, 1.0, mtx.const_data(), v_info,
                                 x.const_data(), v_info, 1.0, work64.data(),
                                 threads);
        };
        auto run_st = [&](unsigned threads) {
            work_st.copy_from(st_res);
            cpu_acc_gemv<ar_type>(m_info, ar_type{1}, st_mtx.const_data(),
                                  v_info, st_x.const_data(), v_info, ar_type{1},
                                  work_st.data(), threads);
        };
        const double t64_1 = time_cpu_ms([&] { run64(1); });
        const double t64_n = time_cpu_ms([&] { run64(cores); });
        const double tst_1 = time_cpu_ms([&] { run_st(1); });
        const double tst_n = time_cpu_ms([&] { run_st(cores); });
        double error{};
        const double t_check = time_cpu_ms([&] {
            error = compare(v_info, work64.const_data(), work_st.const_data(),
                            tmp.data());
        });
        Memory<double> norm_tmp(work64);
        const double norm = reduce<double>(
            v_info, norm_tmp.data(),
            [](double a, double b) { return std::abs(a) + std::abs(b); });
        std::cout << n << DELIM << t64_1 << DELIM << t64_n << DELIM << tst_1
                  << DELIM << tst_n << DELIM << t_check << DELIM
                  << error / norm << '\n';
    }
    return 0;
}


}  // namespace


int main(int argc, char **argv)
{
    constexpr size_type default_max_size{24500};
    constexpr size_type min_size{100};
    cli_options opt;
    const cli_texts texts{
        "compute errors of the GEMVs",
        "set the maximum size (used for both rows and cols) of the matrix.",
        "benchmark different GEMVs", "The matrix size needs to be at least "};
    const int exit_code =
        parse_cli(argc, argv, opt, default_max_size, min_size, texts);
    if (exit_code >= 0) {
        return exit_code;
    }
    return dispatch_types(opt, [&](auto ar, auto st) {
        using ar_type = decltype(ar);
        using st_type = decltype(st);
        return opt.cpu_baseline ? run_cpu_baseline<ar_type, st_type>(opt)
                                : run_gpu<ar_type, st_type>(opt);
    });
}
