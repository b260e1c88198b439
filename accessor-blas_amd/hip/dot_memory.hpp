// Host + device operands of the DOT benchmark. Same contract as `DotMemory<T>`
// of `/root/reference/cuda/dot_memory.cuh`: the generating constructor
// (:36-49) draws x then y, sets the host result to -999 and uploads all three;
// the converting constructor (:61-75) and `convert_from` (:107-112)
// static_cast from another DotMemory; `get_result` (:82-86) fetches the device
// scalar.
#pragma once

#include <type_traits>

#include "matrix_helper.hpp"
#include "memory.hpp"
#include "utils.hpp"


template <typename ValueType>
class DotMemory {
    using Mem = Memory<ValueType>;
    static constexpr auto on_cpu = Mem::Device::cpu;
    static constexpr auto on_gpu = Mem::Device::gpu;

public:
    template <typename VectDist, typename RndEngine>
    DotMemory(matrix_info::size_type size, VectDist &&vect_dist,
              RndEngine &&engine)
        : x_info_{{{size, 1}}},
          y_info_{{{size, 1}}},
          cpu_x_(gen_mtx<ValueType>(x_info_, vect_dist, engine)),
          cpu_y_(gen_mtx<ValueType>(y_info_, vect_dist, engine)),
          cpu_res_(on_cpu, 1),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_y_(on_gpu, y_info_.get_1d_size()),
          gpu_res_(on_gpu, 1)
    {
        *cpu_res_.data() = static_cast<ValueType>(-999);
        copy_cpu_to_gpu();
    }

    template <typename OtherType>
    DotMemory(const DotMemory<OtherType> &other)
        : x_info_(other.x_info_),
          y_info_(other.y_info_),
          cpu_x_(on_cpu, x_info_.get_1d_size()),
          cpu_y_(on_cpu, y_info_.get_1d_size()),
          cpu_res_(on_cpu, 1),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_y_(on_gpu, y_info_.get_1d_size()),
          gpu_res_(on_gpu, 1)
    {
        convert_from(other);
    }

    /** The device scalar, fetched to the host. */
    ValueType get_result()
    {
        cpu_res_.copy_from(gpu_res_);
        return *cpu_res_.data();
    }

    void copy_cpu_to_gpu()
    {
        gpu_x_.copy_from(cpu_x_);
        gpu_y_.copy_from(cpu_y_);
        gpu_res_.copy_from(cpu_res_);
    }

    /** static_cast of x, y and the result from `other`, then upload. */
    template <typename OtherType>
    void convert_from(const DotMemory<OtherType> &other)
    {
        const auto cast = [](OtherType v) { return static_cast<ValueType>(v); };
        convert_mtx(x_info_, other.cpu_x(), cpu_x_nc(), cast);
        convert_mtx(y_info_, other.cpu_y(), cpu_y_nc(), cast);
        convert_mtx(matrix_info{{{1, 1}}}, other.cpu_res(), cpu_res_nc(), cast);
        copy_cpu_to_gpu();
    }

    // non-const host access for re-randomising the vectors
    ValueType *cpu_x_nc() { return cpu_x_.data(); }
    ValueType *cpu_y_nc() { return cpu_y_.data(); }
    ValueType *cpu_res_nc() { return cpu_res_.data(); }

    const ValueType *cpu_x() const { return cpu_x_.data(); }
    const ValueType *cpu_y() const { return cpu_y_.data(); }
    ValueType *cpu_res() { return cpu_res_.data(); }
    const ValueType *cpu_res() const { return cpu_res_.data(); }

    const ValueType *gpu_x() const { return gpu_x_.data(); }
    const ValueType *gpu_y() const { return gpu_y_.data(); }
    ValueType *gpu_res() { return gpu_res_.data(); }
    const ValueType *gpu_res() const { return gpu_res_.data(); }

    const matrix_info x_info_;
    const matrix_info y_info_;

private:
    Mem cpu_x_;
    Mem cpu_y_;
    Mem cpu_res_;
    Mem gpu_x_;
    Mem gpu_y_;
    Mem gpu_res_;
};
