// Host runtime support of the benchmarks, C++14, compiled by g++ — no HIP
// header is needed: everything device-related goes through the C ABI
// (include/accblas.h).
//
// Same surface as `/root/reference/cuda/utils.cuh`: `matrix_info` (:18-56),
// `ceildiv` (:71-75), the error macro (:80-89, here HIP_CALL), `synchronize`
// (:107), the event RAII type and timer (:113-182, here hip_event / HipTimer),
// `benchmark_function` (:236-262: one warm-up, 10 timed runs, MINIMUM in ms)
// and the CPU error-check path `reduce` / `compare` (:281-332).
#pragma once

#include <algorithm>
#include <array>
#include <cassert>
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>

#include <accblas.h>

#include "float16.hpp"


/** Size and row stride of a row-major matrix; a vector is an (n, 1) matrix
 *  whose stride is its increment. */
struct matrix_info {
    using size_type = std::int64_t;
    const std::array<size_type, 2> size;
    const size_type stride;

    constexpr matrix_info(const std::array<size_type, 2> size_,
                          const size_type stride_)
        : size(size_), stride{stride_}
    {}

    /** Dense: stride = number of columns. */
    constexpr matrix_info(const std::array<size_type, 2> size_)
        : matrix_info{size_, size_[1]}
    {}

    /** Elements spanned in memory, gaps of the stride included. */
    size_type get_1d_size() const { return size[0] * stride; }

    /** Elements of the matrix proper. */
    size_type get_num_elems() const { return size[0] * size[1]; }
};


/** ceil(a / b) for positive operands (plain a / b for a <= 0). */
template <typename ValueType>
constexpr ValueType ceildiv(ValueType a, ValueType b)
{
    return (a <= 0) ? a / b : (a - 1) / b + 1;
}


/** accblas_dtype of a host element type. */
template <typename T>
struct accblas_type_of;
template <>
struct accblas_type_of<double> {
    static constexpr accblas_dtype value = ACCBLAS_F64;
};
template <>
struct accblas_type_of<float> {
    static constexpr accblas_dtype value = ACCBLAS_F32;
};
template <>
struct accblas_type_of<float16> {
    static constexpr accblas_dtype value = ACCBLAS_F16;
};


// Any C-ABI call: non-zero status -> one line on stderr, then an exception
// (the reference's CUDA_CALL contract, utils.cuh:80-89).
#define HIP_CALL(call)                                                        \
    do {                                                                      \
        const int accblas_status_ = (call);                                   \
        if (accblas_status_ != ACCBLAS_OK) {                                  \
            const std::string accblas_msg_ = accblas_last_error_string();     \
            std::cerr << "HIP error in file " << __FILE__                     \
                      << " L:" << __LINE__ << "; Error: " << accblas_msg_     \
                      << " (status " << accblas_status_ << ")\n";             \
            throw std::runtime_error(accblas_msg_);                           \
        }                                                                     \
    } while (false)


/** Waits until the device finished all submitted work. */
inline void synchronize() { HIP_CALL(accblas_device_synchronize()); }


/** RAII HIP event. */
struct hip_event {
public:
    hip_event() { HIP_CALL(accblas_event_create(&ev_)); }
    ~hip_event() { accblas_event_destroy(ev_); }
    hip_event(const hip_event &) = delete;
    hip_event &operator=(const hip_event &) = delete;

    /** Destroys and re-creates the event. */
    void reset()
    {
        HIP_CALL(accblas_event_destroy(ev_));
        HIP_CALL(accblas_event_create(&ev_));
    }

    void *get() { return ev_; }

private:
    void *ev_{};
};


/** Event timer on the default stream. */
class HipTimer {
public:
    void start() { HIP_CALL(accblas_event_record(start_.get(), nullptr)); }

    /** Records the end event and waits for it. */
    void stop()
    {
        HIP_CALL(accblas_event_record(end_.get(), nullptr));
        HIP_CALL(accblas_event_synchronize(end_.get()));
    }

    void reset()
    {
        start_.reset();
        end_.reset();
    }

    /** Milliseconds between start() and stop(). */
    double get_time()
    {
        float time{};
        HIP_CALL(accblas_event_elapsed_ms(&time, start_.get(), end_.get()));
        return time;
    }

private:
    hip_event start_;
    hip_event end_;
};


/**
 * Runs `func` once as warm-up (and synchronises); if `skip`, returns 0.
 * Otherwise times 10 further runs with device events and returns the MINIMUM
 * in ms. The timed region contains every launch `func` makes.
 */
template <typename Callable>
double benchmark_function(Callable func, bool skip = false)
{
    constexpr int bench_iters{10};
    func();
    synchronize();
    if (skip) {
        return {};
    }
    HipTimer timer;
    double result_ms{std::numeric_limits<double>::max()};
    for (int i = 0; i < bench_iters; ++i) {
        timer.start();
        func();
        timer.stop();
        result_ms = std::min(result_ms, timer.get_time());
        timer.reset();
    }
    return result_ms;
}


/**
 * Pairwise-tree reduction of the single-column matrix `tmp` with `op`, in
 * place (tmp is destroyed). The pairing is part of the reference's error
 * metric, so it is kept: element r is combined with element r + half, where
 * half = ceil(len / 2), until two elements are left.
 */
template <typename OutputType, typename InputType, typename ReduceOp>
OutputType reduce(const matrix_info info, InputType *tmp, ReduceOp op)
{
    assert(info.size[1] == 1);
    std::int64_t end = info.size[0];
    for (std::int64_t halfway = ceildiv(info.size[0], std::int64_t{2});
         halfway > 1; halfway = ceildiv(halfway, std::int64_t{2})) {
        for (std::int64_t row = 0; row + halfway < end && row < halfway;
             ++row) {
            tmp[row * info.stride] =
                op(tmp[row * info.stride], tmp[(row + halfway) * info.stride]);
        }
        end = halfway;
    }
    return static_cast<OutputType>(info.size[0] == 1
                                       ? op(tmp[0], InputType{})
                                       : op(tmp[0], tmp[info.stride]));
}


/** sum_i |mtx1[i] - mtx2[i]| in ValueType via `reduce`; tmp is scratch. */
template <typename ReferenceType, typename OtherType, typename ValueType>
ValueType compare(const matrix_info info, const ReferenceType *mtx1,
                  const OtherType *mtx2, ValueType *tmp)
{
    assert(info.size[1] == 1);
    for (matrix_info::size_type row = 0; row < info.size[0]; ++row) {
        const auto idx = row * info.stride;
        const ValueType v1 = static_cast<ValueType>(mtx1[idx]);
        const ValueType v2 = static_cast<ValueType>(mtx2[idx]);
        tmp[idx] = std::abs(v1 - v2);
    }
    return reduce<ValueType>(
        info, tmp, [](ValueType a, ValueType b) { return a + b; });
}
