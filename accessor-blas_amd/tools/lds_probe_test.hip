#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const unsigned long long *src, unsigned long long *dst)
{
    __shared__ __attribute__((aligned(16))) unsigned long long pad[16 * 1024];  // 128 KiB in front
    __shared__ __attribute__((aligned(16))) unsigned long long buf[128];
    const int lane = threadIdx.x;
    pad[lane * 251] = lane;  // (keeps `pad` allocated)
    if (lane == 0) dst[128] = static_cast<std::uint32_t>(reinterpret_cast<std::uintptr_t>((__attribute__((address_space(3))) void *)buf));
    buf[lane] = 0; buf[lane + 64] = 0;
    __syncthreads();
    const unsigned long long *mine = src + 2 * lane;
    const std::uint32_t lds = static_cast<std::uint32_t>(reinterpret_cast<std::uintptr_t>(buf));
    asm volatile("s_mov_b32 m0, %[lds]\n"
                 "s_nop 0\n"
                 "global_load_lds_dwordx4 %[ga], off sc1\n"
                 "s_waitcnt vmcnt(0)\n"
                 :
                 : [lds] "s"(lds), [ga] "v"(mine)
                 : "memory", "m0");
    __syncthreads();
    dst[lane] = buf[lane];
    dst[lane + 64] = buf[lane + 64];
    if (pad[lane * 251] != (unsigned long long)lane) dst[129] = 1;
}
int main()
{
    unsigned long long h[128], *s, *d;
    for (int i = 0; i < 128; ++i) h[i] = 0xabcd000000000000ull + i;
    hipMalloc(&s, sizeof(h)); hipMalloc(&d, sizeof(h) + 16); hipMemset(d, 0, sizeof(h) + 16);
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(s, d);
    unsigned long long o[130];
    hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; ++i) bad += o[i] != h[i];
    printf("lds offset of the buffer %llu, pad damaged %llu; mismatches %d; o[0]=%llx o[1]=%llx o[127]=%llx\n", o[128], o[129], bad, o[0], o[1], o[127]);
    return 0;
}
