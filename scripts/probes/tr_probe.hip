// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__global__ void probe(unsigned short* out, int rowstride_bytes)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int l = threadIdx.x;
    int i = l & 15, g = l >> 4;
    // each lane: row (i>>2) of a 4-row block, 8 bytes at col 4*(i&3); groups stacked along rows (k0 = 4*g)
    unsigned addr = (unsigned)(size_t)lds;  // LDS base (low 32 bits of the generic address are the LDS offset?)
    unsigned off = (4 * g + (i >> 2)) * rowstride_bytes + 8 * (i & 3);
    u32x2 v;
    unsigned base = (unsigned)(uintptr_t)(&lds[0]);
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + off) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main()
{
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int rs : {64, 128}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rs);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("rowstride %d bytes (%d elems)\n", rs, rs / 2);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %4d(r%d,c%d)", h[l * 4 + j], h[l * 4 + j] / (rs / 2), h[l * 4 + j] % (rs / 2));
            printf("\n");
        }
    }
    return 0;
}
