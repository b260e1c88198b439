// dot_benchmark [--error] [--size=N] [additional flags, see --help text]
//
// Drop-in for `/root/reference/cuda/dot_benchmark.cu:20-284`: same flags, same
// inputs (seed 42, U(-1,1), x then y; `--error` re-randomises both vectors 10
// times with write_random), same sizes (10^6, then steps of 2*10^6), same
// ';'-separated output: timing mode prints "n;ms per variant;relative error
// per variant", `--error` prints the median-of-10 errors followed by the raw
// results. Vendor columns are rocBLAS (device pointer mode) instead of cuBLAS.
#include <algorithm>
#include <array>
#include <cmath>
#include <ios>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "benchmark_cli.hpp"
#include "cpu_kernels.hpp"
#include "dot_kernels.hpp"
#include "dot_memory.hpp"
#include "memory.hpp"
#include "utils.hpp"


namespace {


using size_type = matrix_info::size_type;
constexpr char DELIM{';'};


struct dot_column {
    std::string name;
    std::function<void(matrix_info, matrix_info)> run;
    std::function<double()> result;  // fetches the device scalar
    std::size_t storage_bytes;
};


DotMemory<double> &dataset(DotMemory<double> &master,
                           std::unique_ptr<DotMemory<double>> &)
{
    return master;
}

template <typename T>
DotMemory<T> &dataset(DotMemory<double> &master,
                      std::unique_ptr<DotMemory<T>> &owned)
{
    owned.reset(new DotMemory<T>(master));
    return *owned;
}

template <typename T>
std::function<double()> result_of(DotMemory<T> &data)
{
    return [&data]() { return static_cast<double>(data.get_result()); };
}

template <typename T>
void add_plain(std::vector<dot_column> &table, myBlasHandle *handle,
               DotMemory<T> &data)
{
    table.push_back(dot_column{
        std::string("DOT ") + type_name<T>::get(),
        [handle, &data](matrix_info x_info, matrix_info y_info) {
            dot(handle, x_info, data.gpu_x(), y_info, data.gpu_y(),
                data.gpu_res());
        },
        result_of(data), sizeof(T)});
}

void add_plain(std::vector<dot_column> &, myBlasHandle *, DotMemory<float16> &)
{}

/** Result type = storage type, as the reference (the holder's scalar is St);
 *  fp16 storage writes an ArType scalar instead so that the result itself is
 *  not rounded to 11 bits. */
template <typename Ar, typename St>
void add_acc(std::vector<dot_column> &table, myBlasHandle *handle,
             DotMemory<St> &data)
{
    table.push_back(dot_column{
        std::string("DOT Acc<") + type_name<Ar>::get() + ", " +
            type_name<St>::get() + ">",
        [handle, &data](matrix_info x_info, matrix_info y_info) {
            acc_dot<Ar>(handle, x_info, data.gpu_x(), y_info, data.gpu_y(),
                        data.gpu_res());
        },
        result_of(data), sizeof(St)});
}

template <typename Ar>
void add_acc_half(std::vector<dot_column> &table, myBlasHandle *handle,
                  DotMemory<float16> &data,
                  std::shared_ptr<Memory<Ar>> gpu_result)
{
    table.push_back(dot_column{
        std::string("DOT Acc<") + type_name<Ar>::get() + ", fp16>",
        [handle, &data, gpu_result](matrix_info x_info, matrix_info y_info) {
            acc_dot<Ar>(handle, x_info, data.gpu_x(), y_info, data.gpu_y(),
                        gpu_result->data());
        },
        [gpu_result]() {
            Memory<Ar> host(Memory<Ar>::Device::cpu, 1);
            host.copy_from(*gpu_result);
            return static_cast<double>(*host.data());
        },
        sizeof(float16)});
}

template <typename Ar, typename St>
void add_reduced(std::vector<dot_column> &table, myBlasHandle *handle,
                 DotMemory<St> &data)
{
    add_acc<Ar, St>(table, handle, data);
    add_acc<St, St>(table, handle, data);
}

template <typename Ar>
void add_reduced(std::vector<dot_column> &table, myBlasHandle *handle,
                 DotMemory<float16> &data)
{
    add_acc_half<Ar>(table, handle, data,
                     std::make_shared<Memory<Ar>>(Memory<Ar>::Device::gpu, 1));
}

template <typename T>
void add_vendor(std::vector<dot_column> &table, accblas_vendor_handle *vendor,
                DotMemory<T> &data)
{
    table.push_back(dot_column{
        std::string("ROCBLAS DOT ") + type_name<T>::get(),
        [vendor, &data](matrix_info x_info, matrix_info y_info) {
            rocblas_dot(vendor, x_info, data.gpu_x(), y_info, data.gpu_y(),
                        data.gpu_res());
        },
        result_of(data), sizeof(T)});
}

void add_vendor(std::vector<dot_column> &, accblas_vendor_handle *,
                DotMemory<float16> &)
{}


template <typename ar_type, typename st_type>
int run_gpu(const cli_options &opt)
{
    constexpr size_type min_size{1000000};
    const bool detailed_error = opt.error_mode;
    const size_type max_size = opt.max_size;

    std::default_random_engine rengine(42);
    std::uniform_real_distribution<double> vector_dist(-1.0, 1.0);

    DotMemory<double> master(max_size, vector_dist, rengine);
    std::unique_ptr<DotMemory<ar_type>> ar_owned;
    std::unique_ptr<DotMemory<st_type>> st_owned;
    DotMemory<ar_type> &ar_data = dataset(master, ar_owned);
    constexpr bool same_types = std::is_same<ar_type, st_type>::value;
    DotMemory<st_type> &st_data =
        same_types ? reinterpret_cast<DotMemory<st_type> &>(ar_data)
                   : *(st_owned.reset(new DotMemory<st_type>(master)),
                       st_owned.get());

    accblas_vendor_handle *vendor{};
    HIP_CALL(accblas_vendor_create(&vendor));
    auto my_handle = std::make_unique<myBlasHandle>();

    // Column order of the reference for <fp64, fp32>: plain ar, plain st,
    // Acc<ar, ar>, Acc<ar, st>, Acc<st, st>, vendor ar, vendor st. Entry 0
    // (fp64 plain kernel) is the error reference.
    std::vector<dot_column> table;
    constexpr size_type benchmark_reference{0};
    add_plain(table, my_handle.get(), master);
    if (!std::is_same<ar_type, double>::value) {
        add_plain(table, my_handle.get(), ar_data);
    }
    if (!same_types) {
        add_plain(table, my_handle.get(), st_data);
    }
    add_acc<ar_type, ar_type>(table, my_handle.get(), ar_data);
    if (!same_types) {
        add_reduced<ar_type>(table, my_handle.get(), st_data);
    }
    add_vendor(table, vendor, master);
    if (!std::is_same<ar_type, double>::value) {
        add_vendor(table, vendor, ar_data);
    }
    if (!same_types) {
        add_vendor(table, vendor, st_data);
    }
    const size_type benchmark_num = static_cast<size_type>(table.size());

    std::cout << "Vector Size";
    if (!detailed_error) {
        for (const auto &entry : table) {
            std::cout << DELIM << entry.name;
        }
    }
    for (const auto &entry : table) {
        std::cout << DELIM << "Error " << entry.name;
    }
    if (opt.rates && !detailed_error) {
        for (const char *what : {"GFLOP/s ", "GB/s ", "%HBM "}) {
            for (const auto &entry : table) {
                std::cout << DELIM << what << entry.name;
            }
        }
    }
    std::cout << '\n';
    std::cout.precision(16);
    std::cout << std::scientific;

    auto get_error = [](double res, double ref_res) -> double {
        return std::abs(res - ref_res) / std::abs(ref_res);
    };

    // reference sweep: 10^6, then +2*10^6 while <= max
    constexpr size_type row_incr = 2000000;
    const auto sizes =
        sweep_sizes(opt, std::min(max_size, min_size), row_incr);
    const size_type num_sizes = static_cast<size_type>(sizes.size());
    constexpr size_type max_randomize_num{10};
    const size_type randomize_num = detailed_error ? max_randomize_num : 1;

    std::vector<double> benchmark_time(num_sizes * benchmark_num);
    std::vector<double> raw_result(randomize_num * num_sizes * benchmark_num);
    const auto raw_idx = [&](size_type rnd, size_type step, size_type bi) {
        return (step * benchmark_num + bi) * randomize_num + rnd;
    };

    for (size_type randomize = 0; randomize < randomize_num; ++randomize) {
        if (randomize != 0) {
            write_random(matrix_info{{{max_size, 1}}}, vector_dist, rengine,
                         master.cpu_x_nc());
            write_random(matrix_info{{{max_size, 1}}}, vector_dist, rengine,
                         master.cpu_y_nc());
            master.copy_cpu_to_gpu();
            if (ar_owned) {
                ar_owned->convert_from(master);
            }
            if (st_owned) {
                st_owned->convert_from(master);
            }
        }
        for (size_type i = 0; i < num_sizes; ++i) {
            const matrix_info x_info{{{sizes[i], 1}}};
            const matrix_info y_info{{{sizes[i], 1}}};
            for (size_type bi = 0; bi < benchmark_num; ++bi) {
                auto curr = [&]() { table[bi].run(x_info, y_info); };
                benchmark_time[i * benchmark_num + bi] =
                    benchmark_function(curr, detailed_error);
                raw_result[raw_idx(randomize, i, bi)] = table[bi].result();
            }
        }
    }

    for (size_type i = 0; i < num_sizes; ++i) {
        std::cout << sizes[i];
        if (!detailed_error) {
            for (size_type bi = 0; bi < benchmark_num; ++bi) {
                std::cout << DELIM << benchmark_time[i * benchmark_num + bi];
            }
            const auto result_ref = raw_result[raw_idx(0, i, benchmark_reference)];
            for (size_type bi = 0; bi < benchmark_num; ++bi) {
                std::cout << DELIM
                          << get_error(raw_result[raw_idx(0, i, bi)], result_ref);
            }
            if (opt.rates) {
                // conventions: 2 n flop; 2 s n bytes
                const double n = static_cast<double>(sizes[i]);
                for (int what = 0; what < 3; ++what) {
                    for (size_type bi = 0; bi < benchmark_num; ++bi) {
                        const double seconds =
                            benchmark_time[i * benchmark_num + bi] * 1e-3;
                        const double gflops = 2.0 * n / seconds / 1e9;
                        const double gbs =
                            2.0 * static_cast<double>(table[bi].storage_bytes) *
                            n / seconds / 1e9;
                        std::cout << DELIM
                                  << (what == 0   ? gflops
                                      : what == 1 ? gbs
                                                  : 100.0 * gbs /
                                                        hbm_peak_gb_per_s);
                    }
                }
            }
        } else {
            for (size_type bi = 0; bi < benchmark_num; ++bi) {
                std::array<double, max_randomize_num> local_error{};
                for (size_type rnd = 0; rnd < randomize_num; ++rnd) {
                    local_error[rnd] =
                        get_error(raw_result[raw_idx(rnd, i, bi)],
                                  raw_result[raw_idx(rnd, i, benchmark_reference)]);
                }
                std::sort(local_error.begin(),
                          local_error.begin() + randomize_num);
                const double median =
                    randomize_num % 2 == 1
                        ? local_error[randomize_num / 2]
                        : (local_error[randomize_num / 2 - 1] +
                           local_error[randomize_num / 2]) /
                              2.0;
                std::cout << DELIM << median;
            }
        }
        std::cout << '\n';
    }
    if (detailed_error) {
        // raw results of every run
        std::cout << "--------------------------------------------------\n";
        std::cout << "Random iter" << DELIM << "Vector Size";
        for (const auto &entry : table) {
            std::cout << DELIM << "Result " << entry.name;
        }
        std::cout << '\n';
        for (size_type i = 0; i < num_sizes; ++i) {
            for (size_type rnd = 0; rnd < randomize_num; ++rnd) {
                std::cout << rnd << DELIM << sizes[i];
                for (size_type bi = 0; bi < benchmark_num; ++bi) {
                    std::cout << DELIM << raw_result[raw_idx(rnd, i, bi)];
                }
                std::cout << '\n';
            }
        }
    }
    my_handle.reset();
    accblas_vendor_destroy(vendor);
    return 0;
}


/** Host-only path (BASELINE.json config 1: dot n=1M fp64/fp64 on the host CPU
 *  reference path, no GPU): the same inputs and sweep through the CPU accessor
 *  kernels, on 1 thread and on all hardware threads. */
template <typename ar_type, typename st_type>
int run_cpu_baseline(const cli_options &opt)
{
    constexpr size_type min_size{1000000};
    const size_type max_size = opt.max_size;
    std::default_random_engine rengine(42);
    std::uniform_real_distribution<double> vector_dist(-1.0, 1.0);
    const matrix_info full{{{max_size, 1}}};
    auto x = gen_mtx<double>(full, vector_dist, rengine);
    auto y = gen_mtx<double>(full, vector_dist, rengine);
    Memory<st_type> st_x(Memory<st_type>::Device::cpu, max_size);
    Memory<st_type> st_y(Memory<st_type>::Device::cpu, max_size);
    const auto cast = [](double v) { return static_cast<st_type>(v); };
    convert_mtx(full, x.const_data(), st_x.data(), cast);
    convert_mtx(full, y.const_data(), st_y.data(), cast);
    const unsigned cores = std::max(1u, std::thread::hardware_concurrency());
    const std::string acc_name = std::string("Acc<") +
                                 type_name<ar_type>::get() + ", " +
                                 type_name<st_type>::get() + ">";
    std::cout << "Vector Size" << DELIM << "CPU DOT fp64 1 core" << DELIM
              << "CPU DOT fp64 " << cores << " cores" << DELIM << "CPU DOT "
              << acc_name << " 1 core" << DELIM << "CPU DOT " << acc_name << ' '
              << cores << " cores" << DELIM << "Result CPU DOT fp64" << DELIM
              << "Result CPU DOT " << acc_name << DELIM << "Error CPU DOT "
              << acc_name << '\n';
    std::cout.precision(16);
    std::cout << std::scientific;
    for (const size_type n :
         sweep_sizes(opt, std::min(max_size, min_size), 2000000)) {
        const matrix_info info{{{n, 1}}};
        double r64{}, r64_mt{};
        ar_type rst{}, rst_mt{};
        const double t64_1 = time_cpu_ms([&] {
            cpu_acc_dot<double>(info, x.const_data(), info, y.const_data(), &r64, 1);
        });
        const double t64_n = time_cpu_ms([&] {
            cpu_acc_dot<double>(info, x.const_data(), info, y.const_data(),
                                &r64_mt, cores);
        });
        const double tst_1 = time_cpu_ms([&] {
            cpu_acc_dot<ar_type>(info, st_x.const_data(), info,
                                 st_y.const_data(), &rst, 1);
        });
        const double tst_n = time_cpu_ms([&] {
            cpu_acc_dot<ar_type>(info, st_x.const_data(), info,
                                 st_y.const_data(), &rst_mt, cores);
        });
        std::cout << n << DELIM << t64_1 << DELIM << t64_n << DELIM << tst_1
                  << DELIM << tst_n << DELIM << r64 << DELIM
                  << static_cast<double>(rst) << DELIM
                  << std::abs(static_cast<double>(rst) - r64) / std::abs(r64)
                  << '\n';
    }
    return 0;
}


}  // namespace


int main(int argc, char **argv)
{
    constexpr size_type min_size{1000000};
    constexpr size_type default_max_size{535 * 1000 * 1000};
    cli_options opt;
    const cli_texts texts{"compute detailed error of the DOTs",
                          "set the maximum size of a vector.",
                          "benchmark different DOTs",
                          "The vector size needs to be at least "};
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
