// RAII buffer on the host or on the current GPU; HIP allocations go through
// the C ABI. Same surface and behaviour as `Memory<T>` of
// `/root/reference/cuda/memory.cuh:14-139`: `copy_from` requires equal element
// counts and throws otherwise (:101-103), copies work in all four
// host/device directions, the copy constructor deep-copies, and releasing a
// device buffer first waits for the device (:60-63).
#pragma once

#include <cstddef>
#include <cstring>
#include <stdexcept>

#include "utils.hpp"


template <typename T>
class Memory {
public:
    enum class Device { cpu, gpu };

    Memory(Device device, std::size_t num_elems)
        : device_{device}, num_elems_{num_elems}, data_{nullptr}
    {
        if (device_ == Device::cpu) {
            data_ = new T[num_elems_];
        } else {
            void *ptr{};
            HIP_CALL(accblas_malloc(&ptr, num_elems_ * sizeof(T)));
            data_ = static_cast<T *>(ptr);
        }
    }

    Memory(const Memory &other) : Memory(other.device_, other.num_elems_)
    {
        *this = other;
    }

    Memory(Memory &&other) noexcept
        : device_{other.device_},
          num_elems_{other.num_elems_},
          data_{other.data_}
    {
        other.data_ = nullptr;
    }

    ~Memory()
    {
        if (data_ == nullptr) {
            return;
        }
        if (device_ == Device::cpu) {
            delete[] data_;
        } else {
            accblas_device_synchronize();
            accblas_free(data_);
        }
        data_ = nullptr;
    }

    Device get_device() const { return device_; }

    std::size_t get_num_elems() const { return num_elems_; }

    T *data() { return data_; }

    const T *data() const { return data_; }

    const T *const_data() const { return data_; }

    /** Copies all elements of `other`, wherever the two buffers live. */
    void copy_from(const Memory &other)
    {
        if (num_elems_ != other.num_elems_) {
            throw std::runtime_error("Mismatching number of elements");
        }
        const std::size_t size_bytes = num_elems_ * sizeof(T);
        const bool to_gpu = device_ == Device::gpu;
        const bool from_gpu = other.device_ == Device::gpu;
        if (!to_gpu && !from_gpu) {
            std::memcpy(data_, other.data_, size_bytes);
        } else if (!to_gpu && from_gpu) {
            HIP_CALL(accblas_memcpy_d2h(data_, other.data_, size_bytes));
        } else if (to_gpu && !from_gpu) {
            HIP_CALL(accblas_memcpy_h2d(data_, other.data_, size_bytes));
        } else {
            HIP_CALL(accblas_memcpy_d2d(data_, other.data_, size_bytes));
        }
    }

    Memory &operator=(const Memory &other)
    {
        this->copy_from(other);
        return *this;
    }

protected:
    const Device device_;
    const std::size_t num_elems_;
    T *data_;
};
