// Developer tool: on which XCD does the workgroup holding ticket t run? The
// 128-row TRSV's "trsv_wide_xcd_run" experiment assumes ticket mod 8.
// usage: xcd_ticket_probe [workgroups=512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void draw(unsigned *counter, unsigned *xcc_of_ticket,
                                            unsigned *cu_of_ticket)
{
    // as much LDS as the solve kernel takes: one workgroup per CU
    __shared__ double occupy[16 * 1024];
    __shared__ unsigned ticket;
    if (threadIdx.x == 0) {
        ticket = atomicAdd(counter, 1u);
        unsigned xcc = 0, hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc_of_ticket[ticket] = xcc & 0xf;
        cu_of_ticket[ticket] = hw;
    }
    __syncthreads();
    occupy[threadIdx.x] = ticket;
    // stay for a while, as a solve's workgroup does
    const auto start = wall_clock64();
    while (wall_clock64() - start < 20000) {
    }
    if (occupy[(threadIdx.x + 1) % 512] < 0) {
        xcc_of_ticket[0] = 99;
    }
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? std::atoi(argv[1]) : 512;
    unsigned *counter, *xcc, *cu;
    hipMalloc(&counter, 4);
    hipMalloc(&xcc, 4 * wgs);
    hipMalloc(&cu, 4 * wgs);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(counter, 0, 4);
        draw<<<wgs, 512>>>(counter, xcc, cu);
        hipDeviceSynchronize();
        std::vector<unsigned> h(wgs);
        hipMemcpy(h.data(), xcc, 4 * wgs, hipMemcpyDeviceToHost);
        int match = 0, match256 = 0;
        for (int t = 0; t < wgs; ++t) {
            match += h[t] == unsigned(t % 8);
            if (t < 256) match256 += h[t] == unsigned(t % 8);
        }
        std::printf("launch %d: xcc == ticket %% 8 for %d of %d tickets (%d of the first 256); first 48:", rep,
                    match, wgs, match256);
        for (int t = 0; t < 48 && t < wgs; ++t) std::printf(" %u", h[t]);
        std::printf("\n");
    }
    return 0;
}
