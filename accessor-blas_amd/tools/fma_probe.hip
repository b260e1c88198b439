// Developer microbenchmark: issue cost of fp64 / fp32 FMA on gfx950 for ONE
// wave with 1..8 independent chains, and with a v_readlane-fed operand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(call) do { hipError_t e = (call); if (e != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); std::exit(1);} } while (0)

template <typename T, int Chains, bool ReadLane>
__global__ void probe(unsigned long long *out, int iters, T *sink)
{
    T acc[Chains];
    for (int c = 0; c < Chains; ++c) acc[c] = threadIdx.x + c;
    T mul = T(1.0000001) + T(threadIdx.x) * T(1e-9);
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < Chains; ++c) {
            T m = mul;
            if (ReadLane) {
                if (sizeof(T) == 8) {
                    int h[2]; __builtin_memcpy(h, &mul, 8);
                    h[0] = __builtin_amdgcn_readlane(h[0], c);
                    h[1] = __builtin_amdgcn_readlane(h[1], c);
                    __builtin_memcpy(&m, h, 8);
                } else {
                    int h; __builtin_memcpy(&h, &mul, 4);
                    h = __builtin_amdgcn_readlane(h, c);
                    __builtin_memcpy(&m, &h, 4);
                }
            }
            acc[c] = acc[c] * m + T(0.5);
        }
    }
    const unsigned long long c1 = clock64();
    T total = 0;
    for (int c = 0; c < Chains; ++c) total += acc[c];
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (total == T(12345.678)) *sink = total;
}

template <typename T, int Chains, bool ReadLane>
void run(const char *name, int threads)
{
    unsigned long long *out; T *sink;
    CHECK(hipMalloc(&out, 8 * 1024)); CHECK(hipMalloc(&sink, 8));
    const int iters = 20000;
    probe<T, Chains, ReadLane><<<1, threads>>>(out, iters, sink);
    probe<T, Chains, ReadLane><<<1, threads>>>(out, iters, sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long h; CHECK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    std::printf("%-10s chains=%d readlane=%d threads=%4d: %6.1f cycles per FMA\n", name, Chains, (int)ReadLane,
                threads, double(h) / iters / Chains);
    CHECK(hipFree(out)); CHECK(hipFree(sink));
}

int main()
{
    run<double, 1, false>("fp64", 64);
    run<double, 2, false>("fp64", 64);
    run<double, 4, false>("fp64", 64);
    run<double, 8, false>("fp64", 64);
    run<double, 8, true>("fp64", 64);
    run<double, 8, false>("fp64", 256);
    run<double, 8, false>("fp64", 512);
    run<double, 8, false>("fp64", 1024);
    run<float, 1, false>("fp32", 64);
    run<float, 8, false>("fp32", 64);
    run<float, 8, true>("fp32", 64);
    return 0;
}
