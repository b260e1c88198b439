// Host + device operands of the TRSV benchmark. Same contract as
// `TrsvMemory<T>` of `/root/reference/cuda/trsv_memory.cuh`: the generating
// constructor (:110-169) draws A then x, keeps a copy of the initial x, then
// factorises the DEVICE matrix in place with a partial-pivot LU that reads the
// buffer column-major and drops the pivots, and copies the factors back to the
// host; that is what makes the triangular systems well conditioned (the
// row-major UPPER triangle then holds the bounded unit factor). The converting
// constructor (:181-197) static_casts the FACTORED matrix and x.
// Differences, both deliberate: the LU is rocSOLVER's `dgetrf` behind
// accblas_vendor_getrf, and its `info` is checked instead of ignored.
#pragma once

#include <stdexcept>
#include <string>
#include <type_traits>

#include <accblas_vendor.h>

#include "matrix_helper.hpp"
#include "memory.hpp"
#include "utils.hpp"


template <typename ValueType>
class TrsvMemory {
    using Mem = Memory<ValueType>;
    static constexpr auto on_cpu = Mem::Device::cpu;
    static constexpr auto on_gpu = Mem::Device::gpu;

public:
    /**
     * cpu_mtx_gen(matrix_info) / cpu_vect_gen(matrix_info) return host
     * Memory<ValueType> objects (the benchmark passes gen_mtx lambdas).
     */
    template <typename MtxGen, typename VectGen>
    TrsvMemory(matrix_info::size_type max_size, MtxGen &&cpu_mtx_gen,
               VectGen &&cpu_vect_gen)
        : m_info_{{{max_size, max_size}}},
          x_info_{{{max_size, 1}}},
          cpu_mtx_(cpu_mtx_gen(m_info_)),
          cpu_x_(cpu_vect_gen(x_info_)),
          cpu_x_init_(cpu_x_),
          gpu_mtx_(on_gpu, m_info_.get_1d_size()),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_x_init_(on_gpu, x_info_.get_1d_size())
    {
        static_assert(std::is_same<ValueType, double>::value,
                      "the factorisation is the fp64 LAPACK routine; generate "
                      "in double and convert");
        upload();
        accblas_vendor_handle *vendor{};
        HIP_CALL(accblas_vendor_create(&vendor));
        int info{};
        const int status = accblas_vendor_getrf(
            vendor, m_info_.size[0], gpu_mtx_.data(), m_info_.stride, &info);
        accblas_vendor_destroy(vendor);
        HIP_CALL(status);
        if (info != 0) {
            throw std::runtime_error("LU factorisation failed, info = " +
                                     std::to_string(info));
        }
        synchronize();
        cpu_mtx_ = gpu_mtx_;
    }

    template <typename OtherType>
    TrsvMemory(const TrsvMemory<OtherType> &other)
        : m_info_(other.m_info_),
          x_info_(other.x_info_),
          cpu_mtx_(on_cpu, m_info_.get_1d_size()),
          cpu_x_(on_cpu, x_info_.get_1d_size()),
          cpu_x_init_(on_cpu, x_info_.get_1d_size()),
          gpu_mtx_(on_gpu, m_info_.get_1d_size()),
          gpu_x_(on_gpu, x_info_.get_1d_size()),
          gpu_x_init_(on_gpu, x_info_.get_1d_size())
    {
        const auto cast = [](OtherType v) { return static_cast<ValueType>(v); };
        convert_mtx(m_info_, other.cpu_mtx_const(), cpu_mtx_.data(), cast);
        convert_mtx(x_info_, other.cpu_x_const(), cpu_x_.data(), cast);
        cpu_x_init_.copy_from(cpu_x_);
        upload();
    }

    /**
     * Transposes the square matrix in place on the host and uploads it again.
     * After the LU the bounded unit factor sits in the row-major upper
     * triangle; a benchmark of the LOWER solve calls this once so that the
     * same well-conditioned factor is the lower triangle. Not in the
     * reference (its system is fixed to upper at compile time).
     */
    void transpose_matrix()
    {
        ValueType *mtx = cpu_mtx_.data();
        const auto n = m_info_.size[0];
        const auto ld = m_info_.stride;
        for (matrix_info::size_type i = 0; i < n; ++i) {
            for (matrix_info::size_type j = i + 1; j < n; ++j) {
                std::swap(mtx[i * ld + j], mtx[j * ld + i]);
            }
        }
        gpu_mtx_.copy_from(cpu_mtx_);
    }

    /** Device x -> host x. */
    void sync_x() { cpu_x_.copy_from(gpu_x_); }

    /** Restores the initial right-hand side on host and device. */
    void reset_x()
    {
        cpu_x_.copy_from(cpu_x_init_);
        gpu_x_.copy_from(gpu_x_init_);
    }

    ValueType *cpu_x() { return cpu_x_.data(); }
    Mem &cpu_x_memory() { return cpu_x_; }
    const ValueType *cpu_mtx_const() const { return cpu_mtx_.const_data(); }
    const ValueType *cpu_x_const() const { return cpu_x_.const_data(); }
    const ValueType *cpu_x_init_const() const
    {
        return cpu_x_init_.const_data();
    }

    ValueType *gpu_x() { return gpu_x_.data(); }
    Mem &gpu_x_memory() { return gpu_x_; }
    const ValueType *gpu_mtx_const() const { return gpu_mtx_.const_data(); }
    const ValueType *gpu_x_const() const { return gpu_x_.const_data(); }

    const matrix_info m_info_;
    const matrix_info x_info_;

private:
    void upload()
    {
        gpu_mtx_.copy_from(cpu_mtx_);
        gpu_x_.copy_from(cpu_x_);
        gpu_x_init_.copy_from(cpu_x_init_);
    }

    Mem cpu_mtx_;
    Mem cpu_x_;
    Mem cpu_x_init_;
    Mem gpu_mtx_;
    Mem gpu_x_;
    Mem gpu_x_init_;
};
