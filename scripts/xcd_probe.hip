// Which XCD does workgroup b of a launch run on?  (test infrastructure)  The RDF pair kernel pins a frame's items to
// the blocks with blockIdx.x % 8 == frame % 8 on the assumption XCD = blockIdx.x % 8; this prints how many blocks
// of a launch honour it, for grids that fill the chip exactly, over-fill it, and under-fill it, at two block sizes.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_probe scripts/xcd_probe.hip && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *xcc, unsigned spin)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0)
        xcc[blockIdx.x] = id & 0xf;
    // keep the block resident for a while so that the grid really fills the chip
    unsigned long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

int main()
{
    for (int threads : {256, 1024})
        for (int blocks : {8 * 32, 8 * 224, 8 * 224 + 8 * 57, 13, 8 * 500}) {
            unsigned *d;
            hipMalloc(&d, sizeof(unsigned) * blocks);
            hipMemset(d, 0xff, sizeof(unsigned) * blocks);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, 200000u);
            hipDeviceSynchronize();
            std::vector<unsigned> h(blocks);
            hipMemcpy(h.data(), d, sizeof(unsigned) * blocks, hipMemcpyDeviceToHost);
            int ok = 0, hist[16] = {0};
            for (int b = 0; b < blocks; ++b) {
                ok += (h[b] == unsigned(b % 8));
                hist[h[b] & 15]++;
            }
            printf("threads %4d blocks %5d: %5d of them on XCD blockIdx %% 8; per XCD:", threads, blocks, ok);
            for (int x = 0; x < 8; ++x)
                printf(" %d", hist[x]);
            printf("  first 16:");
            for (int b = 0; b < 16 && b < blocks; ++b)
                printf(" %u", h[b]);
            printf("\n");
            hipFree(d);
        }
    return 0;
}
