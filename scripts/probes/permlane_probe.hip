// semantics of v_permlane32_swap_b32 on gfx950: r = swap(a, b)  ->  which lanes of r[0], r[1] come from where?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[threadIdx.x] = r[0]; p[threadIdx.x + 64] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r0: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
    printf("r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
