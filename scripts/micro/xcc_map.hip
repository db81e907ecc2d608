// Which XCD does workgroup L of a 1-D grid run on?  (HW_REG_XCC_ID, gfx940+ hwreg 20, bits 0..3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t *out)
{
    if (threadIdx.x == 0) {
        const uint32_t x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        out[blockIdx.x] = x;
    }
}
int main()
{
    const int G = 4096;
    uint32_t *d; hipMalloc(&d, G * 4);
    k<<<G, 512>>>(d);
    std::vector<uint32_t> h(G);
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    int agree = 0; int hist[8][16] = {};
    for (int L = 0; L < G; L++) { hist[L & 7][h[L] & 15]++; }
    for (int r = 0; r < 8; r++) { printf("L%%8=%d:", r); for (int x = 0; x < 8; x++) printf(" xcc%d:%d", x, hist[r][x]); printf("\n"); }
    printf("first 24 ids:"); for (int L = 0; L < 24; L++) printf(" %u", h[L]); printf("\n");
    return 0;
}
