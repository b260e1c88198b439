// Developer probe: issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
// on the current GPU (cycles per instruction for one wave, dependent chain vs
// four independent accumulators, and with two waves on one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

using d4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
using f4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

template <int Chains>
__global__ void probe_f64(double *out, long long *cycles, int iters)
{
    d4 acc[Chains] = {};
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < Chains; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < Chains; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int Chains>
__global__ void probe_f32(float *out, long long *cycles, int iters)
{
    f4 acc[Chains] = {};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < Chains; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < Chains; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    double *out; float *outf; long long *cyc;
    hipMalloc(&out, 1 << 16); hipMalloc(&outf, 1 << 16); hipMalloc(&cyc, 64);
    const int iters = 2000;
    long long h;
    auto report = [&](const char *name, int per_iter, int threads) {
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        std::printf("%-44s %7.1f cycles per MFMA (wave 0; %d threads in the workgroup)\n", name,
                    double(h) / (double(iters) * per_iter), threads);
    };
    probe_f64<1><<<1, 64>>>(out, cyc, iters); report("f64 16x16x4, dependent chain", 1, 64);
    probe_f64<4><<<1, 64>>>(out, cyc, iters); report("f64 16x16x4, 4 independent accumulators", 4, 64);
    probe_f64<4><<<1, 512>>>(out, cyc, iters); report("f64 16x16x4, 4 accumulators, 2 waves per SIMD", 4, 512);
    probe_f32<1><<<1, 64>>>(outf, cyc, iters); report("f32 16x16x4, dependent chain", 1, 64);
    probe_f32<4><<<1, 64>>>(outf, cyc, iters); report("f32 16x16x4, 4 independent accumulators", 4, 64);
    probe_f32<4><<<1, 512>>>(outf, cyc, iters); report("f32 16x16x4, 4 accumulators, 2 waves per SIMD", 4, 512);
    return 0;
}
