// Host-side input generation and conversion; defines the benchmark inputs
// exactly as `/root/reference/cuda/matrix_helper.cuh` does: `gen_mtx` (:28-51)
// fills row-major, leaves stride gaps untouched and redraws until the value is
// a normal number; `write_random` (:66-76) overwrites without that filter;
// `convert_mtx` (:91-103) applies a conversion per element; `print_mtx`
// (:116-133) prints with 7 fixed digits and explicit sign.
#pragma once

#include <cmath>
#include <ios>
#include <iostream>
#include <stdexcept>

#include "memory.hpp"
#include "utils.hpp"


template <typename ValueType, typename ValueDist, typename Engine>
Memory<ValueType> gen_mtx(const matrix_info &info, ValueDist &&dist,
                          Engine &&engine)
{
    if (info.stride < info.size[1]) {
        throw std::runtime_error("Wrong use of stride");
    }
    Memory<ValueType> res(Memory<ValueType>::Device::cpu, info.get_1d_size());
    ValueType *ptr = res.data();
    for (matrix_info::size_type row = 0; row < info.size[0]; ++row) {
        ValueType *row_ptr = ptr + row * info.stride;
        for (matrix_info::size_type col = 0; col < info.size[1]; ++col) {
            ValueType val{};
            do {
                val = dist(engine);
            } while (!std::isnormal(val));
            row_ptr[col] = val;
        }
    }
    return res;
}


template <typename ValueType, typename ValueDist, typename Engine>
void write_random(const matrix_info &info, ValueDist &&dist, Engine &&engine,
                  ValueType *out)
{
    for (matrix_info::size_type row = 0; row < info.size[0]; ++row) {
        for (matrix_info::size_type col = 0; col < info.size[1]; ++col) {
            out[row * info.stride + col] = dist(engine);
        }
    }
}


template <typename ResultType, typename InputType, typename Callable>
void convert_mtx(const matrix_info &info, const InputType *input,
                 ResultType *output, Callable convert)
{
    for (matrix_info::size_type row = 0; row < info.size[0]; ++row) {
        for (matrix_info::size_type col = 0; col < info.size[1]; ++col) {
            const auto idx = row * info.stride + col;
            output[idx] = convert(input[idx]);
        }
    }
}


template <typename ValueType>
void print_mtx(const matrix_info &info, const ValueType *mtx)
{
    const auto saved_flags = std::cout.flags();
    const auto saved_precision = std::cout.precision();
    std::cout.precision(7);
    std::cout << std::fixed << std::showpos;
    for (matrix_info::size_type i = 0; i < info.size[0]; ++i) {
        for (matrix_info::size_type j = 0; j < info.size[1]; ++j) {
            std::cout << static_cast<double>(mtx[i * info.stride + j]) << '\t';
        }
        std::cout << '\n';
    }
    std::cout.flags(saved_flags);
    std::cout.precision(saved_precision);
}
