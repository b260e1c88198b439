// GPU self-test of csrc/wave_primitives.hpp (the gfx950 replacements of the
// reference's `cuda/kernel_utils.cuh` and `cuda/atomics.cuh`): 64-lane
// reductions, workgroup reduction, native and compare-and-swap atomic_add,
// agent-scope store/load helpers and the TRSV read_lane broadcast. Built by the
// Makefile, run by tests/test_gpu_primitives.py. Prints "primitives ok" and
// exits 0 when every check holds.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "trsv_kernels.hpp"
#include "wave_primitives.hpp"

#define CHECK(call)                                                        \
    do {                                                                   \
        hipError_t e = (call);                                             \
        if (e != hipSuccess) {                                             \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,         \
                         hipGetErrorString(e));                            \
            std::exit(2);                                                  \
        }                                                                  \
    } while (0)

using namespace accblas::dev;

// an 8-byte type without a native atomic: exercises the CAS fallback
struct fixed_point {
    long long raw;
    __host__ __device__ fixed_point operator+(fixed_point o) const
    {
        return fixed_point{raw + o.raw};
    }
};

struct results {
    double wave_sum[4];        // per wave of a 256-thread block
    double wave_all[256];      // group_all_reduce<64> result seen by every lane
    double group16[256];       // group_all_reduce<16>
    double block_sum;          // block_reduce_sum<4>
    float atomic_f32;          // 256 * 256 adds of 0.25
    double atomic_f64;
    unsigned int atomic_u32;
    fixed_point atomic_cas;    // CAS fallback
    double lane_broadcast[64];
    double agent_roundtrip;
    // transpose_reduce_sum<Count, Width>: one result per lane
    double transpose_8_32[64];
    double transpose_4_16[64];
    double transpose_16_64[64];
    double transpose_1_64[64];
    float transpose_2_16[64];
    // wave_reduce_sum against the __shfl_down tree it replaces, rounding included
    double tree_new_f64[4], tree_old_f64[4];
    float tree_new_f32[4], tree_old_f32[4];
    double transpose_1_8[64];
    double transpose_4_32[64];
};

__global__ __launch_bounds__(256) void reductions(results *out)
{
    __shared__ double scratch[4];
    const double value = threadIdx.x + 1;  // 1..256
    const double w = wave_reduce_sum(value);
    if (lane_id() == 0) {
        out->wave_sum[wave_id()] = w;
    }
    out->wave_all[threadIdx.x] =
        group_all_reduce<64>(value, [](double a, double b) { return a + b; });
    out->group16[threadIdx.x] =
        group_all_reduce<16>(value, [](double a, double b) { return a + b; });
    {
        // values whose sum depends on the order of the additions
        const double rough = 1.0 / (3.0 + threadIdx.x) + 1e-9 * threadIdx.x * threadIdx.x;
        const float rough_f = static_cast<float>(rough) * 1.000123f;
        double old_d = rough;
        float old_f = rough_f;
        for (int offset = 32; offset > 0; offset >>= 1) {
            old_d += __shfl_down(old_d, offset, 64);
            old_f += __shfl_down(old_f, offset, 64);
        }
        const double new_d = wave_reduce_sum(rough);
        const float new_f = wave_reduce_sum(rough_f);
        if (lane_id() == 0) {
            out->tree_new_f64[wave_id()] = new_d;
            out->tree_old_f64[wave_id()] = old_d;
            out->tree_new_f32[wave_id()] = new_f;
            out->tree_old_f32[wave_id()] = old_f;
        }
    }
    const double total = block_reduce_sum<4>(value, scratch);
    if (threadIdx.x == 0) {
        out->block_sum = total;
        store_agent(&out->agent_roundtrip, 3.25);
        drain_vmem();
        out->agent_roundtrip = load_agent(&out->agent_roundtrip) * 2;
    }
    if (threadIdx.x < 64) {
        // lane l holds l * 1.5; every lane reads lane (63 - l)'s value in turn
        const double mine = threadIdx.x * 1.5;
        double acc = 0;
        for (int src = 0; src < 64; ++src) {
            const double v = kernel::detail::read_lane(mine, src);
            if (src == 63 - (int)threadIdx.x) {
                acc = v;
            }
        }
        out->lane_broadcast[threadIdx.x] = acc;
    }
}

// lane l, value index i carries (i + 1) * 1000 + l (exact in fp32/fp64): the
// group total of index i is Width * (i + 1) * 1000 + sum of the group's lanes
template <int Count, int Width, typename T>
__device__ T transpose_case(int lane)
{
    T values[Count];
    for (int i = 0; i < Count; ++i) {
        values[i] = T((i + 1) * 1000 + lane);
    }
    return transpose_reduce_sum<Count, Width>(values, lane);
}

__global__ __launch_bounds__(64) void transposes(results *out)
{
    const int lane = lane_id();
    out->transpose_8_32[lane] = transpose_case<8, 32, double>(lane);
    out->transpose_4_16[lane] = transpose_case<4, 16, double>(lane);
    out->transpose_16_64[lane] = transpose_case<16, 64, double>(lane);
    out->transpose_1_64[lane] = transpose_case<1, 64, double>(lane);
    out->transpose_2_16[lane] = transpose_case<2, 16, float>(lane);
    out->transpose_1_8[lane] = transpose_case<1, 8, double>(lane);
    out->transpose_4_32[lane] = transpose_case<4, 32, double>(lane);
}

__global__ __launch_bounds__(256) void atomics(results *out)
{
    atomic_add(&out->atomic_f32, 0.25f);
    atomic_add(&out->atomic_f64, 0.5);
    atomic_add(&out->atomic_u32, 1u);
}

// compare-and-swap loops on ONE address serialise badly (every failed attempt
// is another round trip), so the fallback is exercised by one lane per wave
__global__ __launch_bounds__(256) void atomics_cas(results *out)
{
    if (lane_id() == 0) {
        atomic_add(&out->atomic_cas, fixed_point{3});
    }
}

int main()
{
    results *dev;
    CHECK(hipMalloc(&dev, sizeof(results)));
    CHECK(hipMemset(dev, 0, sizeof(results)));
    reductions<<<1, 256>>>(dev);
    transposes<<<1, 64>>>(dev);
    atomics<<<256, 256>>>(dev);
    atomics_cas<<<64, 256>>>(dev);
    CHECK(hipDeviceSynchronize());
    results host;
    CHECK(hipMemcpy(&host, dev, sizeof(results), hipMemcpyDeviceToHost));
    int failures = 0;
    auto expect = [&](bool ok, const char *what) {
        if (!ok) {
            std::printf("FAILED: %s\n", what);
            ++failures;
        }
    };
    for (int w = 0; w < 4; ++w) {
        const double want = 64.0 * (64 * w) + 64.0 * 65 / 2;  // sum of 64w+1 .. 64w+64
        expect(host.wave_sum[w] == want, "wave_reduce_sum");
        expect(host.tree_new_f64[w] == host.tree_old_f64[w] && host.tree_new_f64[w] > 0.0,
               "wave_reduce_sum: the bits of the __shfl_down tree (fp64)");
        expect(host.tree_new_f32[w] == host.tree_old_f32[w] && host.tree_new_f32[w] > 0.0f,
               "wave_reduce_sum: the bits of the __shfl_down tree (fp32)");
        for (int l = 0; l < 64; ++l) {
            expect(host.wave_all[64 * w + l] == want, "group_all_reduce<64>");
        }
    }
    for (int t = 0; t < 256; ++t) {
        const int g = t / 16;
        const double want = 16.0 * (16 * g) + 16.0 * 17 / 2;
        expect(host.group16[t] == want, "group_all_reduce<16>");
    }
    expect(host.block_sum == 256.0 * 257 / 2, "block_reduce_sum");
    const double adds = 256.0 * 256;
    expect(host.atomic_f32 == 0.25f * adds, "atomic_add float");
    expect(host.atomic_f64 == 0.5 * adds, "atomic_add double");
    expect(host.atomic_u32 == (unsigned)adds, "atomic_add unsigned");
    expect(host.atomic_cas.raw == 3 * 64 * 4, "atomic_add CAS fallback");
    for (int l = 0; l < 64; ++l) {
        expect(host.lane_broadcast[l] == (63 - l) * 1.5, "read_lane");
    }
    expect(host.agent_roundtrip == 6.5, "store_agent/load_agent");
    const auto transpose_want = [](int count, int width, int lane) {
        const int index = lane & (count - 1);
        const int first = lane / width * width;
        return double(width) * (index + 1) * 1000 + width * first +
               width * (width - 1) / 2.0;
    };
    for (int l = 0; l < 64; ++l) {
        expect(host.transpose_8_32[l] == transpose_want(8, 32, l), "transpose_reduce_sum<8, 32>");
        expect(host.transpose_4_16[l] == transpose_want(4, 16, l), "transpose_reduce_sum<4, 16>");
        expect(host.transpose_16_64[l] == transpose_want(16, 64, l), "transpose_reduce_sum<16, 64>");
        expect(host.transpose_1_64[l] == transpose_want(1, 64, l), "transpose_reduce_sum<1, 64>");
        expect(host.transpose_2_16[l] == float(transpose_want(2, 16, l)), "transpose_reduce_sum<2, 16>");
        expect(host.transpose_4_32[l] == transpose_want(4, 32, l), "transpose_reduce_sum<4, 32>");
        if (l % 8 == 0) {  // Width 8: lanes [0, Count) of a group only
            expect(host.transpose_1_8[l] == transpose_want(1, 8, l), "transpose_reduce_sum<1, 8>");
        }
    }
    if (failures == 0) {
        std::printf("primitives ok\n");
    }
    return failures == 0 ? 0 : 1;
}
