// Host + device operands of the GEMV benchmark. Same contract as
// `GemvMemory<T>` of `/root/reference/cuda/gemv_memory.cuh`: the generating
// constructor (:39-55) draws A (max x max, row-major), then x, then res — in
// that order — and mirrors them on the GPU; the converting constructor
// (:67-84, :92-108) static_casts every element of another GemvMemory on the
// host and uploads the copies.
#pragma once

#include <type_traits>

#include "matrix_helper.hpp"
#include "memory.hpp"
#include "utils.hpp"


template <typename ValueType>
class GemvMemory {
    using Mem = Memory<ValueType>;
    static constexpr auto on_cpu = Mem::Device::cpu;
    static constexpr auto on_gpu = Mem::Device::gpu;

public:
    /** Random operands of a max_size x max_size problem. */
    template <typename MtxDist, typename VectDist, typename RndEngine>
    GemvMemory(matrix_info::size_type max_size, MtxDist &&mtx_dist,
               VectDist &&vect_dist, RndEngine &&engine)
        : m_info_{{{max_size, max_size}}},
          x_info_{{{max_size, 1}}},
          res_info_{{{max_size, 1}}},
          cpu_mtx_(gen_mtx<ValueType>(m_info_, mtx_dist, engine)),
          cpu_x_(gen_mtx<ValueType>(x_info_, vect_dist, engine)),
          cpu_res_(gen_mtx<ValueType>(res_info_, vect_dist, engine)),
          gpu_mtx_(on_gpu, m_info_.get_1d_size()),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_res_(on_gpu, res_info_.get_1d_size())
    {
        upload();
    }

    /** Element-wise static_cast copy of `other`. */
    template <typename OtherType>
    GemvMemory(const GemvMemory<OtherType> &other)
        : m_info_(other.m_info_),
          x_info_(other.x_info_),
          res_info_(other.res_info_),
          cpu_mtx_(on_cpu, m_info_.get_1d_size()),
          cpu_x_(on_cpu, x_info_.get_1d_size()),
          cpu_res_(on_cpu, res_info_.get_1d_size()),
          gpu_mtx_(on_gpu, m_info_.get_1d_size()),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_res_(on_gpu, res_info_.get_1d_size())
    {
        const auto cast = [](OtherType v) { return static_cast<ValueType>(v); };
        convert_mtx(m_info_, other.cpu_mtx_const(), cpu_mtx_.data(), cast);
        convert_mtx(x_info_, other.cpu_x_const(), cpu_x_.data(), cast);
        convert_mtx(res_info_, other.cpu_res_const(), cpu_res_.data(), cast);
        upload();
    }

    /** Device result -> host result. */
    void sync_result() { cpu_res_.copy_from(gpu_res_); }

    ValueType *cpu_res() { return cpu_res_.data(); }
    Mem &cpu_res_memory() { return cpu_res_; }
    const ValueType *cpu_mtx_const() const { return cpu_mtx_.const_data(); }
    const ValueType *cpu_x_const() const { return cpu_x_.const_data(); }
    const ValueType *cpu_res_const() const { return cpu_res_.const_data(); }
    const ValueType *cpu_mtx() const { return cpu_mtx_const(); }
    const ValueType *cpu_x() const { return cpu_x_const(); }
    const ValueType *cpu_res() const { return cpu_res_const(); }

    ValueType *gpu_res() { return gpu_res_.data(); }
    Mem &gpu_res_memory() { return gpu_res_; }
    const ValueType *gpu_mtx_const() const { return gpu_mtx_.const_data(); }
    const ValueType *gpu_x_const() const { return gpu_x_.const_data(); }
    const ValueType *gpu_res_const() const { return gpu_res_.const_data(); }
    const ValueType *gpu_mtx() const { return gpu_mtx_const(); }
    const ValueType *gpu_x() const { return gpu_x_const(); }
    const ValueType *gpu_res() const { return gpu_res_const(); }

    const matrix_info m_info_;
    const matrix_info x_info_;
    const matrix_info res_info_;

private:
    void upload()
    {
        gpu_mtx_.copy_from(cpu_mtx_);
        gpu_x_.copy_from(cpu_x_);
        gpu_res_.copy_from(cpu_res_);
    }

    Mem cpu_mtx_;
    Mem cpu_x_;
    Mem cpu_res_;
    Mem gpu_mtx_;
    Mem gpu_x_;
    Mem gpu_res_;
};
