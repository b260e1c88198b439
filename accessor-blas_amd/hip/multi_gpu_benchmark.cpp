// multi_gpu_benchmark [--gpus=G] [--dot-size=N] [--gemv-size=N] [--runs=R]
//
// One host process driving G GPUs of one node through include/accblas_comm.h
// (RCCL communicator from ncclCommInitAll, one group per collective). The
// reference has no counterpart (single device, cuda/gemv_benchmark.cu:83-99);
// the two cases are the sharded configurations of SURVEY.md §8e:
//   dot   n (default 2^28), St = fp16, Ar = fp32: G contiguous chunks, one
//         all-reduce of a single fp32 element, checked against a long-double
//         host sum of the same fp16 values;
//   GEMV  n x n (default 32768), St = fp32, Ar = fp64: G row blocks, x
//         replicated, all-gather of the result slices, sampled rows checked
//         against a long-double host product.
// Times are wall-clock minima over R runs between two all-device
// synchronisations, so they include the collective and the launch overhead of
// driving G devices from one thread. Output: one ';'-separated line per case.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include <accblas_comm.h>

#include "dot_kernels.hpp"
#include "float16.hpp"
#include "memory.hpp"
#include "utils.hpp"


namespace {


using size_type = matrix_info::size_type;

#define COMM_CALL(call)                                                      \
    do {                                                                     \
        const int comm_status_ = (call);                                     \
        if (comm_status_ != ACCBLAS_OK) {                                    \
            const std::string comm_msg_ = accblas_comm_last_error_string();  \
            std::cerr << "accblas_comm error in file " << __FILE__           \
                      << " L:" << __LINE__ << "; Error: " << comm_msg_       \
                      << " (status " << comm_status_ << ")\n";               \
            throw std::runtime_error(comm_msg_);                             \
        }                                                                    \
    } while (false)


struct options {
    int gpus{0};  // 0 = every visible device
    size_type dot_size{size_type{1} << 28};
    size_type gemv_size{32768};
    int runs{10};
};

bool parse_value(const std::string &arg, const std::string &flag,
                 long long &out)
{
    if (arg.compare(0, flag.size(), flag) != 0) {
        return false;
    }
    out = std::stoll(arg.substr(flag.size()));
    return true;
}

int parse(int argc, char **argv, options &opt)
{
    for (int i = 1; i < argc; ++i) {
        const std::string arg(argv[i]);
        long long v{};
        if (parse_value(arg, "--gpus=", v)) {
            opt.gpus = static_cast<int>(v);
        } else if (parse_value(arg, "--dot-size=", v)) {
            opt.dot_size = v;
        } else if (parse_value(arg, "--gemv-size=", v)) {
            opt.gemv_size = v;
        } else if (parse_value(arg, "--runs=", v)) {
            opt.runs = static_cast<int>(v);
        } else {
            std::cerr << "Usage: " << argv[0]
                      << " [--gpus=G] [--dot-size=N] [--gemv-size=N]"
                         " [--runs=R]\n"
                         "  sizes of 0 skip the case; G defaults to all "
                         "visible devices\n";
            return 1;
        }
    }
    if (opt.gpus < 0 || opt.dot_size < 0 || opt.gemv_size < 0 || opt.runs < 1) {
        std::cerr << "negative size / gpu count or runs < 1\n";
        return 1;
    }
    return -1;
}


/** U(-1,1) block from the reference's generator (seed 42); buffers larger
 *  than the block repeat it, which keeps host set-up time bounded. */
std::vector<double> random_block(std::size_t count)
{
    std::default_random_engine engine(42);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::vector<double> block(count);
    for (auto &v : block) {
        v = dist(engine);
    }
    return block;
}


/** Everything one rank owns. */
struct rank_state {
    int device{};
    accblas_comm *comm{};
    std::unique_ptr<myBlasHandle> handle;
};


class node {
public:
    explicit node(int gpus)
    {
        int visible{};
        HIP_CALL(accblas_device_count(&visible));
        const int count = gpus == 0 ? visible : gpus;
        if (count < 1 || count > visible) {
            throw std::runtime_error("requested more GPUs than are visible");
        }
        std::vector<accblas_comm *> comms(count);
        COMM_CALL(accblas_comm_create_all(comms.data(), count, nullptr));
        ranks_.resize(count);
        for (int g = 0; g < count; ++g) {
            ranks_[g].comm = comms[g];
            COMM_CALL(accblas_comm_device(comms[g], &ranks_[g].device));
            bind(g);
            ranks_[g].handle.reset(new myBlasHandle());
        }
    }

    ~node()
    {
        for (auto &r : ranks_) {
            accblas_set_device(r.device);
            r.handle.reset();
            accblas_comm_destroy(r.comm);
        }
    }

    int size() const { return static_cast<int>(ranks_.size()); }
    rank_state &operator[](int g) { return ranks_[g]; }
    void bind(int g) { HIP_CALL(accblas_set_device(ranks_[g].device)); }

    void synchronize_all()
    {
        for (int g = 0; g < size(); ++g) {
            bind(g);
            synchronize();
        }
    }

    /** min over `runs` of the wall time of issue() + all-device sync. */
    template <typename Issue>
    double time_ms(int runs, Issue issue)
    {
        issue();  // warm-up, as benchmark_function does
        synchronize_all();
        double best = 1e300;
        for (int r = 0; r < runs; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            issue();
            synchronize_all();
            const auto t1 = std::chrono::steady_clock::now();
            best = std::min(
                best,
                std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        return best;
    }

private:
    std::vector<rank_state> ranks_;
};


void run_dot(node &gpus, size_type n, int runs)
{
    using st_type = float16;
    using ar_type = float;
    const int world = gpus.size();
    const auto block = random_block(std::size_t{1} << 22);
    const auto value_at = [&](size_type i, size_type shift) {
        return st_type(block[static_cast<std::size_t>(i + shift) % block.size()]);
    };
    constexpr size_type y_shift = 1234567;

    std::vector<std::unique_ptr<Memory<st_type>>> x(world), y(world);
    std::vector<std::unique_ptr<Memory<ar_type>>> res(world);
    std::vector<size_type> lo(world), hi(world);
    long double expected = 0;
    for (int g = 0; g < world; ++g) {
        // chunk boundaries on 16-byte vectors of the storage type
        COMM_CALL(accblas_comm_shard(gpus[g].comm, n, 8, &lo[g], &hi[g]));
        const std::size_t len = static_cast<std::size_t>(hi[g] - lo[g]);
        Memory<st_type> hx(Memory<st_type>::Device::cpu, len);
        Memory<st_type> hy(Memory<st_type>::Device::cpu, len);
        long double partial = 0;
        for (std::size_t i = 0; i < len; ++i) {
            hx.data()[i] = value_at(lo[g] + static_cast<size_type>(i), 0);
            hy.data()[i] = value_at(lo[g] + static_cast<size_type>(i), y_shift);
            partial += static_cast<long double>(static_cast<double>(hx.data()[i])) *
                       static_cast<double>(hy.data()[i]);
        }
        expected += partial;
        gpus.bind(g);
        x[g].reset(new Memory<st_type>(Memory<st_type>::Device::gpu, len));
        y[g].reset(new Memory<st_type>(Memory<st_type>::Device::gpu, len));
        res[g].reset(new Memory<ar_type>(Memory<ar_type>::Device::gpu, 1));
        x[g]->copy_from(hx);
        y[g]->copy_from(hy);
    }

    const auto issue = [&]() {
        COMM_CALL(accblas_comm_group_begin());
        for (int g = 0; g < world; ++g) {
            COMM_CALL(accblas_comm_acc_dot(
                gpus[g].comm, gpus[g].handle->get(), ACCBLAS_F32,
                ACCBLAS_F16, ACCBLAS_F32, hi[g] - lo[g], x[g]->data(), 1,
                y[g]->data(), 1, res[g]->data(), nullptr));
        }
        COMM_CALL(accblas_comm_group_end());
    };
    const double ms = gpus.time_ms(runs, issue);

    double worst = 0;
    float first{};
    bool identical = true;
    for (int g = 0; g < world; ++g) {
        gpus.bind(g);
        Memory<ar_type> host(Memory<ar_type>::Device::cpu, 1);
        host.copy_from(*res[g]);
        if (g == 0) {
            first = *host.data();
        }
        identical = identical && *host.data() == first;
        worst = std::max(worst, std::abs(static_cast<double>(
                                    static_cast<long double>(*host.data()) -
                                    expected)));
    }
    const double bytes = 2.0 * sizeof(st_type) * static_cast<double>(n);
    std::cout << "DOT Acc<fp32, fp16>;gpus=" << world << ";n=" << n
              << ";ms=" << ms << ";GB/s=" << bytes / ms * 1e-6
              << ";result=" << first << ";expected=" << static_cast<double>(expected)
              << ";abs_err=" << worst
              << ";all_ranks_identical=" << (identical ? "yes" : "NO") << '\n';
    if (!identical) {
        throw std::runtime_error("ranks disagree on the reduced dot result");
    }
}


void run_gemv(node &gpus, size_type n, int runs)
{
    using st_type = float;
    const int world = gpus.size();
    if (n % world != 0) {
        throw std::runtime_error("--gemv-size must be divisible by the GPU count");
    }
    const size_type rows_local = n / world;
    const auto block = random_block((std::size_t{1} << 22) + 977);
    const auto a_at = [&](size_type row, size_type col) {
        return static_cast<st_type>(
            block[static_cast<std::size_t>(row * 7919 + col) % block.size()]);
    };
    Memory<st_type> host_x(Memory<st_type>::Device::cpu, n);
    for (size_type i = 0; i < n; ++i) {
        host_x.data()[i] = static_cast<st_type>(
            block[static_cast<std::size_t>(i * 31 + 5) % block.size()]);
    }

    std::vector<std::unique_ptr<Memory<st_type>>> a(world), x(world), res(world),
        full(world);
    {
        Memory<st_type> host_a(Memory<st_type>::Device::cpu, rows_local * n);
        for (int g = 0; g < world; ++g) {
            for (size_type r = 0; r < rows_local; ++r) {
                st_type *dst = host_a.data() + r * n;
                for (size_type c = 0; c < n; ++c) {
                    dst[c] = a_at(g * rows_local + r, c);
                }
            }
            gpus.bind(g);
            a[g].reset(new Memory<st_type>(Memory<st_type>::Device::gpu,
                                           rows_local * n));
            x[g].reset(new Memory<st_type>(Memory<st_type>::Device::gpu, n));
            res[g].reset(
                new Memory<st_type>(Memory<st_type>::Device::gpu, rows_local));
            full[g].reset(new Memory<st_type>(Memory<st_type>::Device::gpu, n));
            a[g]->copy_from(host_a);
            x[g]->copy_from(host_x);
        }
    }

    const auto issue_with = [&](bool gather) {
        return [&, gather]() {
            COMM_CALL(accblas_comm_group_begin());
            for (int g = 0; g < world; ++g) {
                COMM_CALL(accblas_comm_acc_gemv(
                    gpus[g].comm, ACCBLAS_F64, ACCBLAS_F32, rows_local, n, 1.0,
                    a[g]->data(), n, x[g]->data(), 1, 0.0, res[g]->data(), 1,
                    gather ? full[g]->data() : nullptr, nullptr));
            }
            COMM_CALL(accblas_comm_group_end());
        };
    };
    const double ms_local = gpus.time_ms(runs, issue_with(false));
    const double ms_gather = gpus.time_ms(runs, issue_with(true));

    // sampled rows against a long-double product; every rank's gathered copy
    // must match rank 0's bit for bit
    Memory<st_type> gathered(Memory<st_type>::Device::cpu, n);
    gpus.bind(0);
    gathered.copy_from(*full[0]);
    double num = 0, den = 0;
    for (size_type row = 0; row < n; row += std::max<size_type>(1, n / 61)) {
        long double sum = 0;
        for (size_type c = 0; c < n; ++c) {
            sum += static_cast<long double>(a_at(row, c)) * host_x.data()[c];
        }
        num += std::abs(static_cast<double>(sum - gathered.data()[row]));
        den += std::abs(static_cast<double>(sum));
    }
    bool identical = true;
    for (int g = 1; g < world; ++g) {
        Memory<st_type> other(Memory<st_type>::Device::cpu, n);
        gpus.bind(g);
        other.copy_from(*full[g]);
        identical = identical &&
                    std::equal(other.data(), other.data() + n, gathered.data());
    }
    const double bytes =
        sizeof(st_type) * (static_cast<double>(n) * n + 2.0 * n);
    std::cout << "GEMV Acc<fp64, fp32>;gpus=" << world << ";n=" << n
              << ";ms=" << ms_local << ";GB/s=" << bytes / ms_local * 1e-6
              << ";ms_with_allgather=" << ms_gather
              << ";rel_err_sampled_rows=" << num / den
              << ";all_ranks_identical=" << (identical ? "yes" : "NO") << '\n';
    if (!identical) {
        throw std::runtime_error("ranks disagree on the gathered GEMV result");
    }
}


}  // namespace


int main(int argc, char **argv)
{
    options opt;
    const int exit_code = parse(argc, argv, opt);
    if (exit_code >= 0) {
        return exit_code;
    }
    try {
        node gpus(opt.gpus);
        std::cout << "Multi-GPU cases on " << gpus.size()
                  << " device(s); wall-clock min of " << opt.runs
                  << " runs incl. collectives\n";
        if (opt.dot_size > 0) {
            run_dot(gpus, opt.dot_size, opt.runs);
        }
        if (opt.gemv_size > 0) {
            run_gemv(gpus, opt.gemv_size, opt.runs);
        }
    } catch (const std::exception &e) {
        std::cerr << "multi_gpu_benchmark: " << e.what() << '\n';
        return 2;
    }
    return 0;
}
