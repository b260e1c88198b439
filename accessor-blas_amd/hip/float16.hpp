// Host-side IEEE binary16 storage type (g++ 11 on x86 has no _Float16).
//
// The reference only instantiates float/double storage
// (`/root/reference/cuda/gemv_benchmark.cu:21-22`); fp16 storage is surface
// this build adds (BASELINE.json config 4). The GPU side uses the native
// `_Float16`; this type has the same bit layout, so buffers cross the C ABI
// unchanged. Conversions are single roundings, round-to-nearest-even, exactly
// what `static_cast<_Float16>(double)` / `(float)` do on the device.
#pragma once

#include <cstdint>
#include <cstring>


struct float16 {
    std::uint16_t bits;

    float16() = default;

    explicit float16(double value) : bits{from_double(value)} {}
    explicit float16(float value)
        : bits{from_double(static_cast<double>(value))}
    {}
    // every fp32 is exactly representable in fp64, so going through double
    // is still a single rounding
    explicit float16(int value) : bits{from_double(value)} {}

    operator float() const { return static_cast<float>(to_double(bits)); }
    operator double() const { return to_double(bits); }

    static float16 from_bits(std::uint16_t b)
    {
        float16 h;
        h.bits = b;
        return h;
    }

    /** fp64 -> fp16, one rounding (RNE), IEEE overflow to infinity. */
    static std::uint16_t from_double(double value)
    {
        std::uint64_t u;
        std::memcpy(&u, &value, sizeof(u));
        const std::uint16_t sign = static_cast<std::uint16_t>((u >> 48) & 0x8000u);
        const int exponent = static_cast<int>((u >> 52) & 0x7ff);
        const std::uint64_t mantissa = u & 0xfffffffffffffull;
        if (exponent == 0x7ff) {  // inf / nan
            if (mantissa == 0) {
                return sign | 0x7c00u;
            }
            // quiet NaN, keep the top payload bits
            return sign | 0x7e00u |
                   static_cast<std::uint16_t>(mantissa >> 42);
        }
        const int e = exponent - 1023;  // unbiased
        if (e >= 16) {
            return sign | 0x7c00u;  // overflow
        }
        if (e >= -14) {
            // normal half: 10 mantissa bits kept, 42 dropped
            std::uint32_t half_exp = static_cast<std::uint32_t>(e + 15);
            std::uint32_t half_man = static_cast<std::uint32_t>(mantissa >> 42);
            const std::uint64_t rest = mantissa & ((1ull << 42) - 1);
            const std::uint64_t halfway = 1ull << 41;
            if (rest > halfway || (rest == halfway && (half_man & 1u))) {
                ++half_man;
                if (half_man == 0x400u) {
                    half_man = 0;
                    ++half_exp;
                }
            }
            if (half_exp >= 31) {
                return sign | 0x7c00u;
            }
            return sign | static_cast<std::uint16_t>((half_exp << 10) | half_man);
        }
        if (e < -25 || exponent == 0) {
            // below half of the smallest half subnormal (2^-24); fp64
            // subnormals are far below that too
            return sign;
        }
        // subnormal half: value = m * 2^-24, m in [0, 1023]
        const std::uint64_t full = mantissa | (1ull << 52);  // 53 bits
        const int shift = 42 + (-14 - e);  // bits dropped; e in [-25, -15]
        const std::uint64_t kept = full >> shift;
        const std::uint64_t rest = full & ((1ull << shift) - 1);
        const std::uint64_t halfway = 1ull << (shift - 1);
        std::uint32_t m = static_cast<std::uint32_t>(kept);
        if (rest > halfway || (rest == halfway && (m & 1u))) {
            ++m;  // may become 0x400 = smallest normal, which is correct
        }
        return sign | static_cast<std::uint16_t>(m);
    }

    static double to_double(std::uint16_t b)
    {
        const std::uint64_t sign = static_cast<std::uint64_t>(b & 0x8000u) << 48;
        const int exponent = (b >> 10) & 0x1f;
        const std::uint64_t mantissa = b & 0x3ffu;
        std::uint64_t u;
        if (exponent == 0x1f) {
            u = sign | 0x7ff0000000000000ull | (mantissa << 42);
        } else if (exponent != 0) {
            u = sign | (static_cast<std::uint64_t>(exponent - 15 + 1023) << 52) |
                (mantissa << 42);
        } else if (mantissa == 0) {
            u = sign;
        } else {
            // subnormal: mantissa * 2^-24
            double magnitude =
                static_cast<double>(mantissa) * 5.9604644775390625e-08;
            std::memcpy(&u, &magnitude, sizeof(u));
            u |= sign;
        }
        double result;
        std::memcpy(&result, &u, sizeof(result));
        return result;
    }
};

static_assert(sizeof(float16) == 2, "float16 must be 2 bytes");
