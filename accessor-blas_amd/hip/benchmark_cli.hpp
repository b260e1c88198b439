// Shared front end of the three benchmark executables: flag parsing, the
// benchmark-table type, CSV helpers and the sweep loops.
//
// The reference repeats this logic in each main
// (`/root/reference/cuda/gemv_benchmark.cu:30-71,127-133,196-250`,
// `cuda/dot_benchmark.cu:30-72,89-94,146-283`, `cuda/trsv_benchmark.cu:32-70,
// 120-125,184-235`); here it is factored out once. The two reference flags are
// kept verbatim (`--error`, `--size=N`, same usage text, same exit codes);
// the flags after them are additions needed for the BASELINE.json
// configurations, which the reference's fixed sweeps cannot hit
// (SURVEY.md §7.3 item 3).
#pragma once

#include <chrono>
#include <cstdint>
#include <functional>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "float16.hpp"
#include "utils.hpp"


struct cli_options {
    using size_type = matrix_info::size_type;
    bool error_mode{false};   // --error
    size_type max_size{};     // --size=N
    // ---- additions -------------------------------------------------------
    size_type only{-1};       // --only=N   run exactly one problem size N
    size_type start{-1};      // --start=N  first size of the sweep
    size_type step{-1};       // --step=N   increment of the sweep
    bool rates{false};        // --rates    append GFLOP/s, GB/s, %HBM columns
    bool cpu_baseline{false}; // --cpu-baseline  host-only path, no GPU touched
    std::string storage{"fp32"};     // --st=fp16|fp32|fp64
    std::string arithmetic{"fp64"};  // --ar=fp32|fp64
};


struct cli_texts {
    std::string error_help;  // e.g. "compute errors of the GEMVs"
    std::string size_help;   // e.g. "set the maximum size (...) of the matrix."
    std::string default_help;  // e.g. "benchmark different GEMVs"
    std::string too_small;   // e.g. "The matrix size needs to be at least "
};


/**
 * Parses argv. Returns -1 to continue, otherwise the process exit code
 * (1 after printing the usage / the minimum-size message, as the reference).
 */
inline int parse_cli(int argc, char **argv, cli_options &opt,
                     cli_options::size_type default_max_size,
                     cli_options::size_type min_size, const cli_texts &texts)
{
    const std::string use_error_string("--error");
    const std::string set_size_string("--size");
    opt.max_size = default_max_size;

    auto print_usage = [&]() {
        const std::string binary(argv[0]);
        std::cerr << "Usage: " << binary << " [" << use_error_string << "] "
                  << '[' << set_size_string << "=SIZE"
                  << "]\n";
        std::cerr << "With:\n"
                  << use_error_string << ":    " << texts.error_help << '\n'
                  << set_size_string << ":     " << texts.size_help
                  << " Default value: " << default_max_size
                  << "; Min value: " << min_size << '\n'
                  << "Without parameters: " << texts.default_help << '\n'
                  << "Additional flags of this build:\n"
                  << "--only=N:   run the single problem size N (row stride = "
                     "max size)\n"
                  << "--start=N, --step=N: first size and increment of the "
                     "sweep\n"
                  << "--st=fp16|fp32|fp64, --ar=fp32|fp64: storage / "
                     "arithmetic type of the reduced-precision columns\n"
                  << "--rates:    append GFLOP/s, GB/s and % of HBM peak "
                     "columns\n"
                  << "--cpu-baseline: time the host-CPU accessor path only "
                     "(no GPU is touched)\n";
    };
    auto value_of = [](const std::string &arg, const std::string &flag) {
        return arg.substr(flag.size() + 1);
    };
    auto starts_with = [](const std::string &arg, const std::string &flag) {
        return arg.substr(0, flag.size()) == flag;
    };

    for (int i = 1; i < argc; ++i) {
        const std::string current(argv[i]);
        if (current == use_error_string) {
            opt.error_mode = true;
        } else if (starts_with(current, set_size_string)) {
            opt.max_size = std::stoll(value_of(current, set_size_string));
        } else if (starts_with(current, "--only=")) {
            opt.only = std::stoll(value_of(current, "--only"));
        } else if (starts_with(current, "--start=")) {
            opt.start = std::stoll(value_of(current, "--start"));
        } else if (starts_with(current, "--step=")) {
            opt.step = std::stoll(value_of(current, "--step"));
        } else if (starts_with(current, "--st=")) {
            opt.storage = value_of(current, "--st");
        } else if (starts_with(current, "--ar=")) {
            opt.arithmetic = value_of(current, "--ar");
        } else if (current == "--rates") {
            opt.rates = true;
        } else if (current == "--cpu-baseline") {
            opt.cpu_baseline = true;
        } else {
            std::cerr << "Unsupported parameter: " << current << '\n';
            print_usage();
            return 1;
        }
    }
    if (opt.only > 0 && opt.only > opt.max_size) {
        opt.max_size = opt.only;
    }
    if (opt.max_size < min_size) {
        std::cerr << texts.too_small << min_size << '\n';
        return 1;
    }
    const bool st_ok = opt.storage == "fp16" || opt.storage == "fp32" ||
                       opt.storage == "fp64";
    const bool ar_ok = opt.arithmetic == "fp32" || opt.arithmetic == "fp64";
    if (!st_ok || !ar_ok) {
        std::cerr << "Unsupported type selection: --ar=" << opt.arithmetic
                  << " --st=" << opt.storage << '\n';
        print_usage();
        return 1;
    }
    return -1;
}


/** The sizes a sweep visits: reference defaults unless --only/--start/--step. */
inline std::vector<cli_options::size_type> sweep_sizes(
    const cli_options &opt, cli_options::size_type default_start,
    cli_options::size_type default_step)
{
    std::vector<cli_options::size_type> sizes;
    if (opt.only > 0) {
        sizes.push_back(opt.only);
        return sizes;
    }
    const auto start = opt.start > 0 ? opt.start : default_start;
    const auto step = opt.step > 0 ? opt.step : default_step;
    for (auto n = start; n <= opt.max_size; n += step) {
        sizes.push_back(n);
    }
    return sizes;
}


template <typename T>
struct type_name;
template <>
struct type_name<double> {
    static const char *get() { return "fp64"; }
};
template <>
struct type_name<float> {
    static const char *get() { return "fp32"; }
};
template <>
struct type_name<float16> {
    static const char *get() { return "fp16"; }
};


/** HBM peak the %-of-roofline columns are quoted against (MI355X, 8 TB/s). */
constexpr double hbm_peak_gb_per_s = 8000.0;


/** One column of a benchmark table. */
template <typename... Infos>
struct benchmark_column {
    std::string name;
    std::function<void(Infos...)> run;       // launches the kernel(s)
    std::function<double(matrix_info)> error;  // error vs reference, resets
    std::size_t storage_bytes;               // bytes per stored element
};


/** Dispatches f.template operator()<Ar, St>() on the --ar / --st strings. */
template <typename F>
int dispatch_types(const cli_options &opt, F &&f)
{
    const bool ar64 = opt.arithmetic == "fp64";
    if (opt.storage == "fp64") {
        return ar64 ? f(double{}, double{}) : f(float{}, double{});
    }
    if (opt.storage == "fp32") {
        return ar64 ? f(double{}, float{}) : f(float{}, float{});
    }
    return ar64 ? f(double{}, float16{}) : f(float{}, float16{});
}


/** min-of-`repeats` wall time of `func` in ms (CPU paths). */
template <typename Callable>
double time_cpu_ms(Callable func, int repeats = 3)
{
    double best = std::numeric_limits<double>::max();
    for (int i = 0; i < repeats; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        func();
        const auto t1 = std::chrono::steady_clock::now();
        best = std::min(
            best, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    return best;
}


/** Runs body(begin, end) over [0, total) split into contiguous blocks, one
 *  std::thread each. */
template <typename Body>
void parallel_blocks(std::int64_t total, unsigned threads, Body body)
{
    if (threads <= 1 || total < 2) {
        body(std::int64_t{0}, total);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        const std::int64_t begin = total * t / threads;
        const std::int64_t end = total * (t + 1) / threads;
        pool.emplace_back([=]() { body(begin, end); });
    }
    for (auto &worker : pool) {
        worker.join();
    }
}
